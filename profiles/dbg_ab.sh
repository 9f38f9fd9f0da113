#!/bin/sh
# TIMING ONLY (results are wrong with any switch set): what do the parts of a pass of ransac_score_prefilter cost?
# reserved[3] = 1024 + mask: 1 no zero-divisor lookups, 2 no append (and no flush), 4 no in-loop flush (ring dropped when full),
# 8 no drain at the end of a pass, 16 no count atomics, 32 no ticket (no wait, no arg-max)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=$O/dbg_ab.txt; : > $OUT
for m in 0 1 2 4 8 16 32 48 12 63 0; do
for h in 1048576 131072; do
python3 bench.py --serial --no-cpu --no-variants --no-extra --hyps $h --steps 60 --reserved 0 0 0 $((1024+m)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mask %2d hyps %7d score %.4f ms clock %4.0f' % ($m, $h, r['avg_launch_ms'], r['shader_clock_mhz']))" >> $OUT
done; done
cat $OUT
