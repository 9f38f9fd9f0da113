#!/bin/sh
# Collects SQ / LDS PMC counters for the RANSAC kernels of bench.py (two passes; no trace domains
# besides --kernel-trace).  Run on the GPU box:  sh profiles/pmc_ransac.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-run}; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $O/pmc_${TAG}_1 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-variants $BENCH_ARGS > $O/pmc_${TAG}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    --output-format csv -d $O/pmc_${TAG}_2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-variants $BENCH_ARGS > $O/pmc_${TAG}_2.log 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_${TAG}_1 $O/pmc_${TAG}_2 > $O/pmc_${TAG}_summary.txt
cat $O/pmc_${TAG}_summary.txt
