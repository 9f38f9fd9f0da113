#!/bin/sh
# Same-box A/B of the working tree's product library against build/old_lib (profiles/build_old_lib.sh <rev>): headline, serial, a rank's share, c3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r06_lib_ab.txt}
: > $OUT
run() {  # label, libdir, args...
  L="$1"; D="$2"; shift; shift
  SFM_AMD_LIB_DIR=$D python3 bench.py --no-cpu --no-variants --no-extra --no-exchange-probe --regions 1 "$@" 2>>$O/r06_lib_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s ms_per_step %.4f score %.4f solve %.4f clock %4.0f kcycles %.1f best %s' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], r['avg_launch_ms']*r['shader_clock_mhz'], d['result']['inliers']))" >> $OUT
}
echo "old = $(cat $R/build/old_lib/REV)" >> $OUT
for rep in 1 2; do
run "new (pipelined)" "" 
run "old (pipelined)" $R/build/old_lib
run "new --serial" "" --serial
run "old --serial" $R/build/old_lib --serial
run "new --hyps 131072" "" --hyps 131072
run "old --hyps 131072" $R/build/old_lib --hyps 131072
run "new c3" "" --config c3
run "old c3" $R/build/old_lib --config c3
done
cat $OUT
