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
# the first configuration (4096 x 2^20): 9 warm-up fills, then rounds of 20 fresh iterations per policy (product's choice, per-hypothesis always,
# per-tile always): one iteration from the first and one from the third run
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfm::", "") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("fill_xu")]
for label, k in (("the product's choice (first call after a fillXU: per-hypothesis operands)", 12), ("per-tile operands for every call (lab bench, reserved[3] = 7)", 9 + 40 + 3)):
    start, stop = idx[k], idx[k + 1]
    t0 = int(rows[start]["Start_Timestamp"])
    prev_end = t0
    print(label)
    for i in range(start, stop):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {names[i][:60]}")
        prev_end = e
PY
rm -rf $O/fresh_trace
