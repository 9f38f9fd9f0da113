R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
TAG=r02b; cfg=headline
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_${TAG}_${cfg}_1 -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_${cfg}_2 -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc_${TAG}_${cfg}_3 -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_${TAG}_${cfg}_1 $O/pmc_${TAG}_${cfg}_2 $O/pmc_${TAG}_${cfg}_3 > $O/pmc_${TAG}_${cfg}_summary.txt
grep -A26 "ransac_score_prefilter" $O/pmc_${TAG}_${cfg}_summary.txt
