#!/bin/sh
# Round-6 evidence in ONE gpurun call (run from the repo root on the GPU box: sh profiles/collect_r06.sh).
# Kernel traces with --kernel-trace --stats; counters in separate --pmc passes with no other trace domain (profiles/pmc_pf.sh).
TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
# 1. counters first (separate passes): headline and c4 -- bench.py quotes the traffic JSON made from them (source hash checked)
sh profiles/pmc_pf.sh ${TAG}_headline > /dev/null 2>&1
sh profiles/pmc_pf.sh ${TAG}_c4 --config c4 > /dev/null 2>&1
sh profiles/pmc_pf.sh ${TAG}_rank8 --hyps 131072 > /dev/null 2>&1
python3 profiles/make_traffic_json.py $O/pmc_${TAG}_headline_summary.txt > $O/${TAG}_traffic.json
cp $O/${TAG}_traffic.json profiles/${TAG}_traffic.json          # (so that the bench lines below quote it; commit the copy)
# 1b. matcher counters on the current kernels (2048^2: exact fp32 MFMA; 16384^2: fp16 pre-filter + exact candidates)
sh profiles/pmc_match.sh 2048 > $O/pmc_${TAG}_match_2048_summary.txt 2>&1
sh profiles/pmc_match.sh 16384 > $O/pmc_${TAG}_match_16384_summary.txt 2>&1
python3 profiles/make_match_traffic_json.py 2048:$O/pmc_${TAG}_match_2048_summary.txt 16384:$O/pmc_${TAG}_match_16384_summary.txt > $O/${TAG}_match_traffic.json
cp $O/${TAG}_match_traffic.json profiles/${TAG}_match_traffic.json
# 2. bench lines: the default command (headline + cpu_baseline + variants + extra configs), serial steps, a rank's share, c3, c4, c5
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
python3 bench.py --serial --no-cpu --no-extra > $O/${TAG}_bench_line_serial.json 2>> $O/${TAG}_bench.err
python3 bench.py --hyps 131072 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_rank8.json 2>> $O/${TAG}_bench.err
python3 bench.py --hyps 131072 --serial --no-cpu --no-variants --no-extra > $O/${TAG}_bench_rank8_serial.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c4 --hyps 131072 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_c4_rank8.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c3 --steps 200 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_c3.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c4 --steps 30 --warmup 5 --no-cpu --no-variants --no-extra > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c5 --steps 20 > $O/${TAG}_bench_c5.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c5 --pairs ring --steps 20 --no-cpu > $O/${TAG}_bench_c5_ring.json 2>> $O/${TAG}_bench.err
# 3. where the time goes inside a scoring launch (lab-bench flavour: block / wavefront stamps), band against G rule
python3 profiles/trace_probe.py > $O/${TAG}_trace.txt 2>/dev/null
python3 profiles/phase_probe.py > $O/${TAG}_phase_probe.txt 2>/dev/null
bash profiles/r06_pack_scan_ab.sh > $O/${TAG}_ab_pack_scan_final.txt 2>&1
# 4. stand-alone benches of the neighbouring rows
MATCH_SIZES=1200,2048,3000,4096,5500,8192,16384 python3 profiles/match_bench.py > $O/${TAG}_match_bench.txt 2>/dev/null
python3 profiles/small_h_bench.py > $O/${TAG}_small_h_bench.txt 2>/dev/null
python3 profiles/fresh_pair_cost.py > $O/${TAG}_fresh_pair_cost.txt 2>/dev/null
sh profiles/fresh_pair_timeline.sh > $O/${TAG}_fresh_pair_timeline.txt 2>&1
{ sh profiles/c1_timeline.sh 1024; sh profiles/c1_timeline.sh 269; } > $O/${TAG}_c1_timeline.txt 2>&1
# 5. rocprof: kernel stats of the bench command (serial steps: one kernel at a time), c3, c4 and a rank's share; pipelined steps
cd /tmp && export TMPDIR=/tmp
for cfg in "headline:" "c3:--config c3" "c4:--config c4 --steps 30 --warmup 5" "rank8:--hyps 131072"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$name -o bench -- python3 $R/bench.py --serial --regions 0 --no-cpu --no-variants --no-extra $args > /dev/null 2>&1
  cp $O/${TAG}_stats_$name/bench_kernel_stats.csv $O/${TAG}_bench_${name}_kernel_stats.csv
  rm -rf $O/${TAG}_stats_$name
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_pipe -o bench -- python3 $R/bench.py --regions 0 --no-cpu --no-variants --no-extra > /dev/null 2>&1
cp $O/${TAG}_stats_pipe/bench_kernel_stats.csv $O/${TAG}_bench_pipelined_kernel_stats.csv; rm -rf $O/${TAG}_stats_pipe
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_c5 -o bench -- python3 $R/bench.py --config c5 --steps 10 --regions 1 --no-cpu > /dev/null 2>&1
cp $O/${TAG}_stats_c5/bench_kernel_stats.csv $O/${TAG}_bench_c5_kernel_stats.csv; rm -rf $O/${TAG}_stats_c5
cd $R
head -c 1200 $O/${TAG}_bench_line.json; echo; head -4 $O/${TAG}_bench_headline_kernel_stats.csv | cut -c1-200; cat $O/${TAG}_traffic.json | head -50
