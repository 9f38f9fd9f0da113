#!/bin/sh
# A/B of the scoring kernel variants in ONE gpurun call (same box): bench.py --serial, headline + rank-8 shard + c4.
# usage: sh profiles/pf_ab.sh [out-file]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/pf_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/pf_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s ms_per_step %.4f score %.4f solve %.4f frac %.3f clock %4.0f grid %d best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['frac'], r['shader_clock_mhz'], d['config']['kernel']['grid'], d['result']))" >> $OUT
}
for rep in 1 2; do
run "r3 default (pipelined steps)"
run "r2 kernel  (pipelined steps)" --reserved 0 0 0 2
run "r3 --serial" --serial
run "r2 --serial" --serial --reserved 0 0 0 2
run "r3 static passes --serial" --serial --reserved 0 2
run "r3 stand-alone prep --serial" --serial --reserved 0 0 0 3
run "r3 --serial --hyps 131072" --serial --hyps 131072
run "r2 --serial --hyps 131072" --serial --hyps 131072 --reserved 0 0 0 2
run "r3 --hyps 131072 (pipelined)" --hyps 131072
run "r2 --hyps 131072 (pipelined)" --hyps 131072 --reserved 0 0 0 2
run "r3 --serial c4" --serial --config c4 --steps 20
run "r2 --serial c4" --serial --config c4 --steps 20 --reserved 0 0 0 2
run "r3 --serial c3" --serial --config c3
run "r2 --serial c3" --serial --config c3 --reserved 0 0 0 2
done
cat $OUT
