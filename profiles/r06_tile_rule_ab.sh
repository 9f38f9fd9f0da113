#!/bin/bash
# Round 6: per-tile band constants (kPfRuleBandTile, the product: 7 = no switch) against the packed scan with per-hypothesis records and whole-view boxes (reserved[3] = 6)
# and round 5's alignbit scan (reserved[3] = 5): headline, serial, a rank's share, c3, c4 -- lab-bench library, one box
run() {
  echo "== $*"
  timeout 300 python bench.py --no-extra --no-variants --no-cpu --no-exchange-probe --regions 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('ms_per_step %.4f score %.4f solve %.4f clock %4.0f kcycles %.1f' % (d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], r['avg_launch_ms']*r['shader_clock_mhz']))
"
}
for rep in 1 2; do
for r in "0 0 0 7" "0 0 0 6" "0 0 0 5"; do
  run --reserved $r
  run --serial --reserved $r
  run --hyps 131072 --reserved $r
  run --config c3 --reserved $r
  run --config c4 --steps 20 --reserved $r
done
done
