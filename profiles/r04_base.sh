#!/bin/sh
# Round-4 baseline on today's box: headline (pipelined + serial), a rank's share, c3 -- one line each.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r04_base.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/r04_base.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s ms_per_step %.4f score %.4f solve %.4f frac %.3f clock %4.0f best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['frac'], r['shader_clock_mhz'], d['result']))" >> $OUT
}
for rep in 1 2; do
run "default (pipelined)"
run "--serial" --serial
run "--hyps 131072 (pipelined)" --hyps 131072
run "--serial --hyps 131072" --serial --hyps 131072
run "--serial c3" --serial --config c3
done
cat $OUT
