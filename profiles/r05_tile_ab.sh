#!/bin/sh
# Round 5: tiles of up to 1536 points (3 x 1376 for 4096 points, 11 x 1504 for 16384) against the 1024-point tiles of rounds 2-4
# (reserved[1] == 6, lab-bench library), band rule, ONE gpurun call (same box).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_tile_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra --regions 1 "$@" 2>>$O/r05_tile_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s ms_per_step %.4f score %.4f solve %.4f clock %4.0f grid %d lds %d best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['config']['kernel']['grid'], d['config']['kernel']['lds_bytes'], d['result']['inliers']))" >> $OUT
}
for rep in 1 2; do
run "big tiles (pipelined)" --reserved 0 7
run "1024    (pipelined)" --reserved 0 6
run "big tiles --serial" --serial --reserved 0 7
run "1024    --serial" --serial --reserved 0 6
run "big tiles --hyps 131072" --hyps 131072 --reserved 0 7
run "1024    --hyps 131072" --hyps 131072 --reserved 0 6
run "big tiles --serial c4" --serial --config c4 --steps 20 --reserved 0 7
run "1024    --serial c4" --serial --config c4 --steps 20 --reserved 0 6
run "big tiles --serial c3" --serial --config c3 --reserved 0 7
run "1024    --serial c3" --serial --config c3 --reserved 0 6
done
cat $OUT
