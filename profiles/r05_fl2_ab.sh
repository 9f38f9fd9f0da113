#!/bin/sh
# Round 5: a 256-entry survivor ring flushed 128 entries at a time, two per lane (reserved[1] == 9) against the 64-entry flush (any other
# value), lab-bench library, band rule, ONE gpurun call (same box).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_fl2_ab.txt}
: > $OUT
run() {  # label, args...
  L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra --regions 1 "$@" 2>>$O/r05_fl2_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s ms_per_step %.4f score %.4f solve %.4f clock %4.0f lds %d best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['config']['kernel']['lds_bytes'], d['result']['inliers']))" >> $OUT
}
for rep in 1 2; do
run "fl2   (pipelined)" --reserved 0 9
run "flush (pipelined)" --reserved 0 10
run "fl2   --serial" --serial --reserved 0 9
run "flush --serial" --serial --reserved 0 10
run "fl2   --hyps 131072" --hyps 131072 --reserved 0 9
run "flush --hyps 131072" --hyps 131072 --reserved 0 10
run "fl2   --serial c4" --serial --config c4 --steps 20 --reserved 0 9
run "flush --serial c4" --serial --config c4 --steps 20 --reserved 0 10
run "fl2   --serial c3" --serial --config c3 --reserved 0 9
run "flush --serial c3" --serial --config c3 --reserved 0 10
done
cat $OUT
