#!/bin/sh
# Round 5: the lane-solve kernel's wavefronts at issue priority 3 / 1 (s_setprio; lab-bench library, reserved[1] == 11 / 12) against the
# default priority 0, in the pipelined steps where the solve of step k + 1 runs next to the scoring of step k.  ONE gpurun call (same box).
# The switches are not in the tree (they would change the hash the committed counters are keyed by): apply profiles/r05_solve_prio_experiment.patch
# and `make` first.
# usage: sh profiles/r05_prio_ab.sh [out-file]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_ab_solve_prio.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/r05_prio_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s ms_per_step %.4f regions %s score %.4f solve %.4f clock %4.0f best %s' % ('$L', d['ms_per_step'], d.get('ms_per_step_regions',{}).get('all'), r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['result'].get('best_hypothesis')))" >> $OUT
}
for rep in 1 2 3; do
run "prio 0 (lab-bench library, inert switch)" --reserved 0 13
run "prio 3" --reserved 0 11
run "prio 1" --reserved 0 12
run "scoring at prio 2, solve at 0" --reserved 0 14
run "prio 0 --hyps 131072" --hyps 131072 --reserved 0 13
run "prio 3 --hyps 131072" --hyps 131072 --reserved 0 11
run "scoring at prio 2 --hyps 131072" --hyps 131072 --reserved 0 14
run "prio 0 c4" --config c4 --steps 20 --reserved 0 13
run "prio 3 c4" --config c4 --steps 20 --reserved 0 11
done
cat $OUT
