#!/bin/sh
# Round 5: band rule (one v_alignbit per pair, two MFMAs per 32 x 32 pairs) against the G rule of rounds 2-4 (reserved[3] == 4, lab-bench
# library), ONE gpurun call (same box): headline, a rank's share, c3, c4; pipelined and serial.
# usage: sh profiles/r05_rule_ab.sh [out-file]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_rule_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/r05_rule_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s ms_per_step %.4f score %.4f solve %.4f clock %4.0f grid %d best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['config']['kernel']['grid'], d['result']))" >> $OUT
}
for rep in 1 2; do
run "band (pipelined steps)"
run "G    (pipelined steps)" --reserved 0 0 0 4
run "band --serial" --serial
run "G    --serial" --serial --reserved 0 0 0 4
run "band --serial --hyps 131072" --serial --hyps 131072
run "G    --serial --hyps 131072" --serial --hyps 131072 --reserved 0 0 0 4
run "band --hyps 131072 (pipelined)" --hyps 131072
run "G    --hyps 131072 (pipelined)" --hyps 131072 --reserved 0 0 0 4
run "band --serial c4" --serial --config c4 --steps 20
run "G    --serial c4" --serial --config c4 --steps 20 --reserved 0 0 0 4
run "band --serial c3" --serial --config c3
run "G    --serial c3" --serial --config c3 --reserved 0 0 0 4
done
cat $OUT
