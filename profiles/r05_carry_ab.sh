#!/bin/sh
# Round 5: the scoring kernel without the end-of-pass drain (leftovers ride into the next pass: ransac_score_prefilter_carry) against the
# one with it (reserved[1] == 8, lab-bench library), band rule, ONE gpurun call (same box).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_carry_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra --regions 1 "$@" 2>>$O/r05_carry_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s ms_per_step %.4f score %.4f solve %.4f clock %4.0f lds %d best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['config']['kernel']['lds_bytes'], d['result']['inliers']))" >> $OUT
}
for rep in 1 2; do
run "carry (pipelined)" --reserved 0 9
run "drain (pipelined)" --reserved 0 8
run "carry --serial" --serial --reserved 0 9
run "drain --serial" --serial --reserved 0 8
run "carry --hyps 131072" --hyps 131072 --reserved 0 9
run "drain --hyps 131072" --hyps 131072 --reserved 0 8
run "carry --serial c4" --serial --config c4 --steps 20 --reserved 0 9
run "drain --serial c4" --serial --config c4 --steps 20 --reserved 0 8
run "carry --serial c3" --serial --config c3 --reserved 0 9
run "drain --serial c3" --serial --config c3 --reserved 0 8
done
cat $OUT
