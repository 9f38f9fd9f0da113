#!/bin/sh
# kernel timeline of fillXU + estimateE on a fresh point set (profiles/fresh_pair_cost.py under rocprofv3 --kernel-trace): where the once-per-fillXU time goes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/fresh_trace
rocprofv3 --kernel-trace --output-format csv -d $O/fresh_trace -o t -- python3 $R/profiles/fresh_pair_cost.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/fresh_trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the first configuration's fresh loop: find a run of fill_xu -> ... -> finalize and print three iterations from the middle
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfm::", "") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("fill_xu")]
start = idx[10] if len(idx) > 12 else 0
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for i in range(start, min(start + 27, len(rows))):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {names[i][:60]}")
    prev_end = e
PY
rm -rf $O/fresh_trace
