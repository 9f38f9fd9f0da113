#!/bin/sh
# Round-2 evidence in ONE gpurun call (run from the repo root on the GPU box: sh profiles/collect_r02.sh).
# Kernel traces with --kernel-trace --stats; counters in separate --pmc passes with no other trace domain.
TAG=r02
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
# 1. bench lines: headline (with cpu_baseline + variants), BASELINE configs[2] / [3], the plain wavefront kernel on each (A/B)
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
python3 bench.py --serial --no-cpu > $O/${TAG}_bench_line_serial.json 2>> $O/${TAG}_bench.err
python3 bench.py --hyps 131072 --no-cpu --no-variants > $O/${TAG}_bench_rank8.json 2>> $O/${TAG}_bench.err
python3 bench.py --hyps 131072 --serial --no-cpu --no-variants > $O/${TAG}_bench_rank8_serial.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c3 --steps 200 --warmup 20 --no-cpu --no-variants > $O/${TAG}_bench_c3.json 2>> $O/${TAG}_bench.err
python3 bench.py --config c4 --steps 30 --warmup 5 --no-cpu --no-variants > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench.err
: > $O/${TAG}_prefilter_ab.txt
for cfg in headline c3 c4; do for k in 4 1; do
  python3 bench.py --serial --config $cfg --steps 30 --warmup 5 --no-cpu --no-variants --kernel $k 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'config':'$cfg','kernel':d['config']['kernel']['name'],'ms_per_step':round(d['ms_per_step'],4),'score_ms':round(r['avg_launch_ms'],4),'frac':round(r['frac'],4),'clock_mhz':round(r['shader_clock_mhz']),'grid':d['config']['kernel']['grid'],'result':d['result']}))" >> $O/${TAG}_prefilter_ab.txt
done; done
for h in 16384 32768 65536 131072 262144; do for k in 4 1; do
  python3 bench.py --serial --hyps $h --steps 50 --warmup 5 --no-cpu --no-variants --kernel $k 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'matches':4096,'hypotheses':$h,'kernel':d['config']['kernel']['name'],'ms_per_step':round(d['ms_per_step'],4),'score_ms':round(r['avg_launch_ms'],4),'solve_ms':round(r['solve_kernel_avg_ms'],4)}))" >> $O/${TAG}_prefilter_ab.txt
done; done
# 2. stand-alone benches
python3 profiles/pipeline_bench.py > $O/${TAG}_pipeline_bench.txt 2>/dev/null
python3 profiles/ring_bench.py > $O/${TAG}_ring_bench.txt 2>/dev/null
MATCH_SIZES=1200,2048,3000,4096,5500,8192,16384 python3 profiles/match_bench.py > $O/${TAG}_match_bench.txt 2>/dev/null
python3 profiles/homography_bench.py > $O/${TAG}_homography_bench.txt 2>/dev/null
python3 profiles/sift_bench.py > $O/${TAG}_sift_bench.txt 2>/dev/null
python3 profiles/small_h_bench.py > $O/${TAG}_small_h_bench.txt 2>/dev/null
./profiles/probes/mfma_f16_probe.bin > $O/${TAG}_mfma_f16_probe.txt 2>&1
./profiles/probes/pk_clamp_probe.bin > $O/${TAG}_pk_clamp_probe.txt 2>&1
python3 tests/fuzz_gpu.py 300 99 > $O/${TAG}_fuzz.txt 2>/dev/null
python3 profiles/prefilter_soak.py 420 31 > $O/${TAG}_prefilter_soak.txt 2>/dev/null
python3 profiles/soak_repro.py "$(cat profiles/soak_cases_r02.json)" $O/${TAG}_soak_repro.npz 2>/dev/null | tail -1 > $O/${TAG}_soak_repro_after_fix.txt
python3 profiles/phase_probe.py 2>/dev/null | grep hypotheses > $O/${TAG}_phase_probe.txt
python3 profiles/enqueue_probe.py 2>/dev/null | grep hypotheses > $O/${TAG}_enqueue_probe.txt
sh profiles/small_shard_ab.sh 2>/dev/null | grep hyps > $O/${TAG}_small_shard_ab.txt
# 3. rocprof: kernel stats of the bench command (headline, c3, c4) and one multi-GPU-sized step (131072 hypotheses: what one of 8 ranks runs)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o bench -- python3 $R/bench.py --serial --steps 100 --warmup 20 --no-cpu --no-variants > /dev/null 2>&1
cp $O/${TAG}_stats/bench_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_c3 -o bench -- python3 $R/bench.py --serial --config c3 --steps 100 --warmup 20 --no-cpu --no-variants > /dev/null 2>&1
cp $O/${TAG}_stats_c3/bench_kernel_stats.csv $O/${TAG}_bench_c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_c4 -o bench -- python3 $R/bench.py --serial --config c4 --steps 30 --warmup 5 --no-cpu --no-variants > /dev/null 2>&1
cp $O/${TAG}_stats_c4/bench_kernel_stats.csv $O/${TAG}_bench_c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_rank -o bench -- python3 $R/bench.py --serial --hyps 131072 --steps 100 --warmup 20 --no-cpu --no-variants > /dev/null 2>&1
cp $O/${TAG}_stats_rank/bench_kernel_stats.csv $O/${TAG}_bench_rank8_kernel_stats.csv
python3 - > $O/${TAG}_rank8_step_timeline.txt <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/${TAG}_stats_rank/bench_kernel_trace.csv")) if 'sfm' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
i0=len(rows)//2
while 'solve' not in rows[i0]['Kernel_Name']: i0+=1
t0=int(rows[i0]['Start_Timestamp'])
print("one estimateE over 131072 hypotheses x 4096 matches (the share of one of 8 ranks), ns from the start of its first kernel")
for r in rows[i0:i0+4]:
    print("%-32s start %7d end %7d" % (r['Kernel_Name'].split('(')[0].replace('void ','')[-32:], int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_pipe -o bench -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu --no-variants > /dev/null 2>&1
cp $O/${TAG}_stats_pipe/bench_kernel_stats.csv $O/${TAG}_bench_pipelined_kernel_stats.csv
# 3b. the matchers: per-kernel durations and counters at 16384 x 16384
sh $R/profiles/match_kernels.sh 5500,16384 > $O/${TAG}_match_kernels.txt 2>/dev/null
sh $R/profiles/pmc_match.sh 16384 > /dev/null 2>&1
grep -A18 "match_pf\|match_mfma" $O/pmc_match_summary.txt > $O/pmc_${TAG}_match_summary.txt
cd /tmp
# 4. counters (separate passes): SQ / LDS, matrix pipe, HBM traffic -- bench command, headline and c4
for cfg in headline c4; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_${TAG}_${cfg}_1 -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_${cfg}_2 -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc_${TAG}_${cfg}_3 -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_${cfg}_f -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_${cfg}_w -o p -- python3 $R/bench.py --serial --config $cfg --steps 3 --warmup 1 --no-cpu --no-variants > /dev/null 2>&1
python3 $R/profiles/pmc_summary.py $O/pmc_${TAG}_${cfg}_1 $O/pmc_${TAG}_${cfg}_2 $O/pmc_${TAG}_${cfg}_3 $O/pmc_${TAG}_${cfg}_f $O/pmc_${TAG}_${cfg}_w > $O/pmc_${TAG}_${cfg}_summary.txt
done
python3 $R/profiles/make_traffic_json.py $O/pmc_${TAG}_headline_summary.txt > $O/${TAG}_traffic.json
cat $O/${TAG}_prefilter_ab.txt | head -8; head -5 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200; cat $O/${TAG}_rank8_step_timeline.txt; cat $O/${TAG}_traffic.json | head -30; tail -3 $O/${TAG}_ring_bench.txt | cut -c1-300; tail -1 $O/${TAG}_fuzz.txt; tail -1 $O/${TAG}_prefilter_soak.txt
