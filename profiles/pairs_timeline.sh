#!/bin/sh
# Kernel timeline of the many-pairs driver (profiles/ring_bench.py): how much of the pairs phase is kernels, how much overlaps.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/pairs_trace -o t -- python3 $R/profiles/ring_bench.py > $O/pairs_trace_bench.txt 2>/dev/null
cd $R
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/pairs_trace/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last dino_all_630 run: take the last 630 pose_chain kernels
idx = [i for i, r in enumerate(rows) if "pose_chain" in r["Kernel_Name"]]
first = idx[-630]
t0 = int(rows[first]["Start_Timestamp"]) - 200000
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
t1 = max(int(r["End_Timestamp"]) for r in sel)
print("window %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(sel)))
d = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:]
    d[k][0] += 1; d[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print("%-46s calls %5d total %8.1f us avg %6.2f us" % (k, v[0], v[1], v[1] / v[0]))
# union of busy intervals and average concurrency
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = 0; cur = 0; last = None; area = 0
for t, dlt in ev:
    if cur > 0: busy += t - last; area += (t - last) * cur
    cur += dlt; last = t
print("some kernel running %.2f ms of the window; mean concurrency while busy %.2f; queues used %s" % (busy / 1e6, area / max(busy, 1), sorted(set(r.get("Queue_Id", "?") for r in sel))))
PY
