R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
for c in 32 64 128 256 512 1024; do
python3 bench.py --steps 30 --warmup 5 --no-cpu --no-variants --kernel 4 --reserved 0 0 $c 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cols $c', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['kernel']['grid'])"
done
BENCH_ARGS="--kernel 4" sh profiles/pmc_ransac.sh r02pf > /dev/null 2>&1
cat $O/pmc_r02pf_summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc_r02pf_3 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-variants --kernel 4 > $O/pmc_r02pf_3.log 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_r02pf_3 | grep -A12 prefilter
