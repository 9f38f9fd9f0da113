#!/bin/sh
# Per-kernel durations of the matchers (rocprofv3 --kernel-trace of profiles/match_bench.py): sh profiles/match_kernels.sh [sizes]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MATCH_NO_CPU=1 MATCH_SIZES=${1:-5500,16384} rocprofv3 --kernel-trace --stats --output-format csv -d $O/mp_stats -o m -- python3 $R/profiles/match_bench.py > $O/mp_bench.txt 2>/dev/null
cd $R
python3 - <<PY
import csv
from collections import defaultdict
rows = [r for r in csv.DictReader(open("$O/mp_stats/m_kernel_trace.csv")) if "match" in r["Kernel_Name"]]
d = defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"].split("(")[0][-44:], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(d.items()):
    print("%-46s grid %-8s calls %3d avg %.1f us min %.1f us" % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
PY
cut -c1-330 $O/mp_bench.txt
