R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
BENCH_ARGS="--kernel 4" sh profiles/pmc_ransac.sh r02pf2 > /dev/null 2>&1
grep -A20 "ransac_score_prefilter" $O/pmc_r02pf2_summary.txt | head -24
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc_r02pf2_3 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-variants --kernel 4 > $O/pmc_r02pf2_3.log 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_r02pf2_3 | grep -A10 prefilter
