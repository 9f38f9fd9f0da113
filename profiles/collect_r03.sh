#!/bin/sh
# Round-3 evidence in ONE gpurun call (run from the repo root on the GPU box: sh profiles/collect_r03.sh).
# Kernel traces with --kernel-trace --stats; counters in separate --pmc passes with no other trace domain (profiles/pmc_pf.sh).
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
# 1. bench lines: the default command (headline + cpu_baseline + variants + extra configs), serial steps, a rank's share
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
python3 bench.py --serial --no-cpu --no-extra > $O/${TAG}_bench_line_serial.json 2>> $O/${TAG}_bench.err
python3 bench.py --hyps 131072 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_rank8.json 2>> $O/${TAG}_bench.err
python3 bench.py --hyps 131072 --serial --no-cpu --no-variants --no-extra > $O/${TAG}_bench_rank8_serial.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c4 --hyps 131072 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_c4_rank8.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c3 --steps 200 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_c3.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c4 --steps 30 --warmup 5 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench.err
# 2. A/B on the same box: round-3 kernel against the round-2 kernel, static pass order, stand-alone prep
sh profiles/pf_ab.sh $O/${TAG}_pf_ab.txt > /dev/null 2>&1
python3 profiles/trace_probe.py > $O/${TAG}_trace.txt 2>/dev/null
# 3. stand-alone benches of the neighbouring rows
python3 profiles/pipeline_bench.py > $O/${TAG}_pipeline_bench.txt 2>/dev/null
python3 profiles/ring_bench.py > $O/${TAG}_ring_bench.txt 2>/dev/null
MATCH_SIZES=1200,2048,3000,4096,5500,8192,16384 python3 profiles/match_bench.py > $O/${TAG}_match_bench.txt 2>/dev/null
python3 profiles/homography_bench.py > $O/${TAG}_homography_bench.txt 2>/dev/null
python3 profiles/sift_bench.py > $O/${TAG}_sift_bench.txt 2>/dev/null
python3 profiles/small_h_bench.py > $O/${TAG}_small_h_bench.txt 2>/dev/null
python3 tests/fuzz_gpu.py 200 77 > $O/${TAG}_fuzz.txt 2>/dev/null
python3 profiles/prefilter_soak.py 300 41 > $O/${TAG}_prefilter_soak.txt 2>/dev/null
# 4. rocprof: kernel stats of the bench command (serial steps: one kernel at a time), c3, c4 and a rank's share
cd /tmp && export TMPDIR=/tmp
for cfg in "headline:" "c3:--config c3" "c4:--config c4 --steps 30 --warmup 5" "rank8:--hyps 131072"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$name -o bench -- python3 $R/bench.py --serial --no-cpu --no-variants --no-extra $args > /dev/null 2>&1
  cp $O/${TAG}_stats_$name/bench_kernel_stats.csv $O/${TAG}_bench_${name}_kernel_stats.csv
  rm -rf $O/${TAG}_stats_$name
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_pipe -o bench -- python3 $R/bench.py --no-cpu --no-variants --no-extra > /dev/null 2>&1
cp $O/${TAG}_stats_pipe/bench_kernel_stats.csv $O/${TAG}_bench_pipelined_kernel_stats.csv; rm -rf $O/${TAG}_stats_pipe
# 5. counters (separate passes): headline, c4, the round-2 kernel on the headline
cd $R
sh profiles/pmc_pf.sh ${TAG}_headline > /dev/null 2>&1
sh profiles/pmc_pf.sh ${TAG}_c4 --config c4 > /dev/null 2>&1
sh profiles/pmc_pf.sh ${TAG}_r2kernel --reserved 0 0 0 2 > /dev/null 2>&1
python3 profiles/make_traffic_json.py $O/pmc_${TAG}_headline_summary.txt > $O/${TAG}_traffic.json
head -c 1500 $O/${TAG}_bench_line.json; echo; head -8 $O/${TAG}_pf_ab.txt; head -3 $O/${TAG}_bench_headline_kernel_stats.csv | cut -c1-220; cat $O/${TAG}_traffic.json | head -40; tail -1 $O/${TAG}_fuzz.txt; tail -1 $O/${TAG}_prefilter_soak.txt
