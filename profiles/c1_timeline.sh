#!/bin/sh
# Kernel timeline of ONE dino pair end to end (profiles/c1_timeline.py): per kernel of an iteration, its duration and the gap
# between the end of the previous kernel and its start -- where the 64 us of BASELINE configs[1] go.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
H=${1:-1024}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/c1_trace -o t -- python3 $R/profiles/c1_timeline.py $H > $O/c1_timeline_run.txt 2>/dev/null
cd $R
tail -1 $O/c1_timeline_run.txt
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/c1_trace/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfm::", "")[-40:]
# an iteration starts at a matcher kernel; take the last 50 iterations
starts = [i for i, r in enumerate(rows) if "match_" in r["Kernel_Name"] and "none" not in r["Kernel_Name"]]
starts = starts[-51:]
per = collections.OrderedDict()
tot = []
for a, b in zip(starts[:-1], starts[1:]):
    it = rows[a:b]
    tot.append((int(rows[b]["Start_Timestamp"]) - int(it[0]["Start_Timestamp"])) / 1e3)
    prev_end = None
    for k, r in enumerate(it):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        key = (k, name(r))
        d = per.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3; d[2] += 0.0 if prev_end is None else (s - prev_end) / 1e3
        prev_end = e
print("iterations %d, start-to-start %.2f us (with the profiler attached)" % (len(tot), sum(tot) / len(tot)))
print("%-3s %-42s %8s %8s" % ("#", "kernel", "dur us", "gap us"))
sd = sg = 0.0
for (k, nm), d in per.items():
    if d[0] < len(tot) // 2: continue
    print("%-3d %-42s %8.2f %8.2f" % (k, nm, d[1] / d[0], d[2] / d[0])); sd += d[1] / d[0]; sg += d[2] / d[0]
print("sum of durations %.2f us, of gaps inside an iteration %.2f us" % (sd, sg))
PY
