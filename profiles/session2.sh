R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
$R/profiles/probes/mfma_f16_probe.bin > $O/r02_mfma_f16_probe.txt 2>&1
cat $O/r02_mfma_f16_probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02a_c3_stats -o c3 -- python3 $R/bench.py --config c3 --steps 100 --warmup 20 --no-cpu --no-variants > $O/r02a_c3_prof.log 2>&1
head -8 $O/r02a_c3_stats/c3_kernel_stats.csv
