#!/bin/sh
# Round 5: the lane-solve kernel's register allocation capped for 5 wavefronts per SIMD (96 registers, 6 spills; run as "product" here)
# against 4 (104 registers, what the compiler picks unconstrained: the product before and after), 6 (80, 46 spills) and 8 (64, 106 spills).
# Negative: the switches (template parameter WPE of ransac_solve_lanes1_qr, reserved[1] = 15 / 16 / 17) were removed again; to re-run,
# make the kernel `template <int WPE> __launch_bounds__(64, WPE)` and select it in launch_ransac_score.  Same box, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_ab_solve_occupancy.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/r05_solve_occupancy_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s ms_per_step %.4f regions %s score %.4f solve %.4f clock %4.0f best %s' % ('$L', d['ms_per_step'], d.get('ms_per_step_regions',{}).get('all'), r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['result'].get('best_hypothesis')))" >> $OUT
}
for rep in 1 2 3; do
run "5 waves/SIMD (product)" --reserved 0 13
run "4 waves/SIMD" --reserved 0 15
run "6 waves/SIMD" --reserved 0 16
run "8 waves/SIMD" --reserved 0 17
run "5 --serial" --serial --reserved 0 13
run "4 --serial" --serial --reserved 0 15
run "6 --serial" --serial --reserved 0 16
run "5 --hyps 131072 --serial" --hyps 131072 --serial --reserved 0 13
run "4 --hyps 131072 --serial" --hyps 131072 --serial --reserved 0 15
run "6 --hyps 131072 --serial" --hyps 131072 --serial --reserved 0 16
run "5 --hyps 131072" --hyps 131072 --reserved 0 13
run "4 --hyps 131072" --hyps 131072 --reserved 0 15
done
cat $OUT
