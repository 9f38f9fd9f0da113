#!/bin/sh
# Round 5: the normal-equations + Jacobi solver (jacobi_sweeps = 7) as two hypotheses per lane (packed, 508 registers: the product's
# choice) against one hypothesis per lane (the generic scalar kernel, lab-bench reserved[0] == 1), same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_jacobi_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra --regions 1 --steps 30 "$@" 2>>$O/r05_jacobi_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s ms_per_step %.4f score %.4f solve %.4f clock %4.0f best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['result']['inliers']))" >> $OUT
}
for rep in 1 2; do
run "jacobi 7, two per lane packed (serial)" --serial --sweeps 7 --reserved 5
run "jacobi 7, one per lane scalar (serial)" --serial --sweeps 7 --reserved 1
run "jacobi 7, two per lane packed (pipelined)" --sweeps 7 --reserved 5
run "jacobi 7, one per lane scalar (pipelined)" --sweeps 7 --reserved 1
done
cat $OUT
