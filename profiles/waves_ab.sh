#!/bin/sh
# 16 against 12 wavefronts per scoring block (reserved[1] = 5), pipelined and serial steps, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/waves_ab.txt}; : > $OUT
run() { L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/waves_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s ms_per_step %.4f score %.4f solve %.4f clock %4.0f block %d' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['config']['kernel']['block']))" >> $OUT
}
for rep in 1 2; do
run "16 waves, pipelined"
run "12 waves, pipelined" --reserved 0 5
run "16 waves, serial" --serial
run "12 waves, serial" --serial --reserved 0 5
run "16 waves, pipelined, 131072" --hyps 131072
run "12 waves, pipelined, 131072" --hyps 131072 --reserved 0 5
run "16 waves, pipelined, c4" --config c4 --steps 20
run "12 waves, pipelined, c4" --config c4 --steps 20 --reserved 0 5
done
cat $OUT
