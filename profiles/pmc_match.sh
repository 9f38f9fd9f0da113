#!/bin/sh
# Counters of the matcher kernels at one size (separate --pmc passes, no other trace domain; FETCH_SIZE / WRITE_SIZE in passes of their own):
# sh profiles/pmc_match.sh [n]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
N=${1:-16384}
cd /tmp && export TMPDIR=/tmp
MATCH_NO_CPU=1 MATCH_SIZES=$N rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_match_1 -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
MATCH_NO_CPU=1 MATCH_SIZES=$N rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc_match_2 -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
MATCH_NO_CPU=1 MATCH_SIZES=$N rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_match_f -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
MATCH_NO_CPU=1 MATCH_SIZES=$N rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_match_w -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_match_1 $O/pmc_match_2 $O/pmc_match_f $O/pmc_match_w > $O/pmc_match_summary.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$O/pmc_match_1/*kernel_trace.csv"):
    from collections import defaultdict
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "match" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        print(k, "avg ns under counters", sum(v) / len(v))
PY
cat $O/pmc_match_summary.txt
