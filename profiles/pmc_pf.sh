#!/bin/sh
# Counters of the scoring kernel for one bench configuration (separate --pmc passes, kernel trace only).
# usage: sh profiles/pmc_pf.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_summary.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --serial --steps 3 --warmup 1 --no-cpu --no-variants --no-extra --no-exchange-probe $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_${TAG}_1 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_2 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_${TAG}_3 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_f -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_w -o p -- $B > /dev/null 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_${TAG}_1 $O/pmc_${TAG}_2 $O/pmc_${TAG}_3 $O/pmc_${TAG}_f $O/pmc_${TAG}_w > $O/pmc_${TAG}_summary.txt
rm -rf $O/pmc_${TAG}_1 $O/pmc_${TAG}_2 $O/pmc_${TAG}_3 $O/pmc_${TAG}_f $O/pmc_${TAG}_w
cat $O/pmc_${TAG}_summary.txt
