#!/bin/bash
# kernel time + a few counters of the fused matcher at one size (N, default 2048); run on the GPU box through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/mfprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o mf -- python3 $R/profiles/match_fused_probe.py > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc1 -o p -- python3 $R/profiles/match_fused_probe.py > /dev/null 2>&1
cd $R
for f in $(find gpurun_out/mfprof/stats -name "*kernel_stats.csv"); do head -6 $f; done
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/mfprof/pmc1/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / max(1, n[(k, c)])) for c, v in d.items()})
PY
