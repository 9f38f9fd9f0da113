#!/bin/sh
# Collects the round's rocprofv3 evidence on the GPU box (run from the repo root: sh profiles/collect.sh r01).
#   1. kernel trace + stats of the default bench command
#   2. SQ / LDS counters of the RANSAC kernels (two passes, profiles/pmc_ransac.sh)
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot limit), bench, matcher and SIFT extractor
#   4. kernel trace + stats of the SIFT extractor (profiles/sift_bench.py)
# Counter passes use --kernel-trace only (no sys/hip/hsa trace domains).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o bench -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu --no-variants > $O/${TAG}_bench_prof.log 2>&1
cp $O/${TAG}_stats/bench_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
export MATCH_NO_CPU=1      # the profiled matcher runs skip the CPU leg
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_mstats -o match -- python3 $R/profiles/match_bench.py > $O/${TAG}_match_prof.log 2>&1
cp $O/${TAG}_mstats/match_kernel_stats.csv $O/${TAG}_match_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_mfetch -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_mwrite -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${TAG}_mpmc -o p -- python3 $R/profiles/match_bench.py > /dev/null 2>&1
SIFT_REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_sstats -o sift -- python3 $R/profiles/sift_bench.py > $O/${TAG}_sift_prof.log 2>&1
cp $O/${TAG}_sstats/sift_kernel_stats.csv $O/${TAG}_sift_kernel_stats.csv
SIFT_REPS=3 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_sfetch -o p -- python3 $R/profiles/sift_bench.py > /dev/null 2>&1
SIFT_REPS=3 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_swrite -o p -- python3 $R/profiles/sift_bench.py > /dev/null 2>&1
python3 $R/profiles/pmc_summary.py $O/${TAG}_sfetch $O/${TAG}_swrite > $O/${TAG}_traffic_sift.txt
python3 $R/profiles/pmc_summary.py $O/${TAG}_fetch $O/${TAG}_write > $O/${TAG}_traffic_bench.txt
python3 $R/profiles/pmc_summary.py $O/${TAG}_mfetch $O/${TAG}_mwrite $O/${TAG}_mpmc > $O/${TAG}_traffic_match.txt
cd $R && sh profiles/pmc_ransac.sh $TAG > /dev/null 2>&1
cat $O/${TAG}_bench_kernel_stats.csv | head -8; cat $O/${TAG}_traffic_bench.txt | head -30; cat $O/${TAG}_match_kernel_stats.csv | head -5; cat $O/${TAG}_traffic_match.txt; cat $O/${TAG}_sift_kernel_stats.csv | head -12; cat $O/${TAG}_traffic_sift.txt | head -30
python3 $R/profiles/sift_timeline.py $O/${TAG}_sstats/sift_kernel_trace.csv > $O/${TAG}_sift_timeline_1080p.json
