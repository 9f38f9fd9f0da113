R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_ring_stats -o ring -- python3 $R/profiles/ring_bench.py > $O/r02_ring_prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/r02_ring_stats/ring_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"].split("(")[0][-50:], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
tail -2 $O/r02_ring_prof.log
