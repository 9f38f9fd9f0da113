"""Sums rocprofv3 counter_collection CSVs per kernel and prints per-launch averages."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(acc):
    if "sfm::" not in k and "3sfm" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        n = max(len(calls[(k, c)]), 1)
        print(f"    {c:28s} {acc[k][c] / n:16.0f} per launch ({n} launches)")
    v = acc[k]
    n = max(len(calls[(k, "SQ_WAVE_CYCLES")]), 1)
    if v.get("SQ_WAVE_CYCLES"):
        print(f"    -> VALU active / wave-cycles {v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']:.3f}, "
              f"wait_any {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}, wait_inst_any {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}, "
              f"cycles per VALU inst {4 * v['SQ_ACTIVE_INST_VALU'] / max(v['SQ_INSTS_VALU'], 1):.2f}")
