#!/bin/sh
# A/B of the packed scan (v_pk_fma_f32, default) against the plain one (reserved[1] == 6) in ONE gpurun call
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/pk_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/pk_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s ms_per_step %.4f score %.4f solve %.4f frac %.3f clock %4.0f best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['frac'], r['shader_clock_mhz'], d['result']))" >> $OUT
}
for rep in 1 2 3; do
run "packed scan (pipelined steps)"
run "plain scan  (pipelined steps)" --reserved 0 6
run "packed scan --serial" --serial
run "plain scan  --serial" --serial --reserved 0 6
run "packed scan --serial c4" --serial --config c4 --steps 20
run "plain scan  --serial c4" --serial --config c4 --steps 20 --reserved 0 6
done
cat $OUT
