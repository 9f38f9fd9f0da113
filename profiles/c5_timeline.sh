#!/bin/sh
# Kernel timeline of the configs[4] job (bench.py --config c5: 36 dino views -> 630 pairs): wall-clock extent of each phase of
# the LAST step (front end / matcher launches / estimateE + poses), busy time per kernel kind and the gaps between phases.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/c5_trace -o t -- python3 $R/bench.py --config c5 --steps 6 --warmup 3 --regions 1 --no-cpu > $O/c5_timeline_run.txt 2>/dev/null
cd $R
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/c5_trace/t_kernel_trace.csv")))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfm::", "")[-36:]
# steps end with triangulate_pairs_kernel; take the last complete step = kernels after the previous triangulate_pairs end
tri = [i for i, r in enumerate(rows) if "triangulate_pairs" in r["Kernel_Name"]]
a, b = tri[-2] + 1, tri[-1] + 1
step = rows[a:b]
t0 = step[0]["s"]
def phase(pred):
    sel = [r for r in step if pred(r["Kernel_Name"])]
    if not sel: return None
    busy = 0; cur = None
    for r in sorted(sel, key=lambda r: r["s"]):
        if cur is None or r["s"] > cur[1]:
            if cur: busy += cur[1] - cur[0]
            cur = [r["s"], r["e"]]
        else: cur[1] = max(cur[1], r["e"])
    busy += cur[1] - cur[0]
    return (min(r["s"] for r in sel) - t0) / 1e3, (max(r["e"] for r in sel) - t0) / 1e3, busy / 1e3, len(sel), sum(r["e"] - r["s"] for r in sel) / 1e3
print("last step: %d kernels, %.1f us from its first kernel to its last" % (len(step), (step[-1]["e"] - t0) / 1e3))
for title, pred in (("front end (u8 -> float, SIFT kernels, copies, fills)", lambda k: "sift" in k or "views_u8" in k or "rocclr" in k),
                    ("matcher launches", lambda k: "match_" in k),
                    ("fill_xu_pairs", lambda k: "fill_xu_pairs" in k),
                    ("ransac_pairs_solve", lambda k: "ransac_pairs_solve" in k),
                    ("ransac_fused_pairs", lambda k: "ransac_fused_pairs" in k),
                    ("choose_pose_pairs", lambda k: "choose_pose_pairs" in k),
                    ("triangulate_pairs", lambda k: "triangulate_pairs" in k)):
    p = phase(pred)
    if p: print("  %-54s from %8.1f to %8.1f us   some kernel running %8.1f us   %4d kernels, %9.1f us of kernel time" % ((title,) + p))
d = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    d[name(r)][0] += 1; d[name(r)][1] += (r["e"] - r["s"]) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:14]:
    print("    %-38s calls %4d total %8.1f us avg %7.2f us" % (k, v[0], v[1], v[1] / v[0]))
print("queues used:", sorted(set(r.get("Queue_Id", "?") for r in step)))
PY
tail -1 $O/c5_timeline_run.txt | cut -c1-300
