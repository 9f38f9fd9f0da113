#!/bin/sh
# Round 5: the normal-equations + Jacobi solver (jacobi_sweeps = 7): one hypothesis per lane with the registers capped at 256 (two
# wavefronts per SIMD: the product's choice; lab-bench library with the inert switch reserved[0] == 5) against two hypotheses per lane
# packed (502 registers; the product up to round 4; reserved[0] == 2), one per lane unconstrained (257 registers; 1) and capped at 168
# (99 spills; 7).  Same box, lab-bench library throughout.
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
run "jacobi 7, two per lane packed (serial)" --serial --sweeps 7 --reserved 2
run "jacobi 7, one per lane scalar (serial)" --serial --sweeps 7 --reserved 1
run "jacobi 7, one per lane, <= 256 regs (serial)" --serial --sweeps 7 --reserved 5
run "jacobi 7, one per lane, <= 168 regs (serial)" --serial --sweeps 7 --reserved 7
run "jacobi 7, two per lane packed (pipelined)" --sweeps 7 --reserved 2
run "jacobi 7, one per lane scalar (pipelined)" --sweeps 7 --reserved 1
run "jacobi 7, one per lane, <= 256 regs (pipelined)" --sweeps 7 --reserved 5
run "jacobi 7, one per lane, <= 168 regs (pipelined)" --sweeps 7 --reserved 7
done
cat $OUT
