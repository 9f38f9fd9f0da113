#!/bin/sh
# Same-box A/B of TWO BUILDS of the library (another commit's libsfm_amd.so under build/old_lib, see SFM_AMD_LIB_DIR in
# cuda-sfm_amd/__init__.py).  usage: sh profiles/r04_ab_libs.sh <out> "<label>|<bench args>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=$1; shift
: > $OUT
for rep in 1 2 3; do
for spec in "$@"; do
  L=${spec%%|*}; A=${spec#*|}
  for lib in new old; do
  if [ $lib = old ]; then export SFM_AMD_LIB_DIR=$R/build/old_lib; else unset SFM_AMD_LIB_DIR; fi
  python3 bench.py --no-cpu --no-variants --no-extra --regions 1 $A 2>>$O/r04_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-4s %-28s ms_per_step %.4f score %.4f solve %.4f clock %4.0f best %s' % ('$lib', '$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['result']['best_hypothesis']))" >> $OUT
  done
done
done
unset SFM_AMD_LIB_DIR
sort -k2,4 -s $OUT
