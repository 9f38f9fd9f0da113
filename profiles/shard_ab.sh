#!/bin/sh
# the share of one of eight ranks (131072 x 4096, and 131072 x 16384): whole-tile against half-tile passes, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/shard_ab.txt}; : > $OUT
run() { L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/shard_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-52s ms_per_step %.4f score %.4f solve %.4f clock %4.0f grid %d' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['config']['kernel']['grid']))" >> $OUT
}
for rep in 1 2; do
run "131072 x 4096 whole-tile passes (pipelined)" --hyps 131072 --reserved 0 3
run "131072 x 4096 half-tile passes  (pipelined)" --hyps 131072 --reserved 0 4
run "131072 x 4096 whole-tile passes --serial" --serial --hyps 131072 --reserved 0 3
run "131072 x 4096 half-tile passes  --serial" --serial --hyps 131072 --reserved 0 4
run "131072 x 16384 whole-tile (pipelined)" --config c4 --hyps 131072 --reserved 0 3
run "131072 x 16384 half-tile  (pipelined)" --config c4 --hyps 131072 --reserved 0 4
run "2^20 x 4096 whole-tile (pipelined)" --reserved 0 3
run "2^20 x 4096 half-tile  (pipelined)" --reserved 0 4
run "65536 x 16384 (c3) whole-tile --serial" --serial --config c3 --reserved 0 3
run "65536 x 16384 (c3) half-tile  --serial" --serial --config c3 --reserved 0 4
done
cat $OUT
