#!/bin/sh
# Round-4 A/B of scoring-kernel arrangements in ONE gpurun call (same box).  usage: sh profiles/r04_ab.sh <out> "<label>|<bench args>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=$1; shift
: > $OUT
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for spec in "$@"; do
  L=${spec%%|*}; A=${spec#*|}
  python3 bench.py --no-cpu --no-variants --no-extra --regions 1 $A 2>>$O/r04_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s ms_per_step %.4f score %.4f solve %.4f frac %.3f clock %4.0f best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['frac'], r['shader_clock_mhz'], d['result']['best_hypothesis']))" >> $OUT
done
done
cat $OUT
