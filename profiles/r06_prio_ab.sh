#!/bin/bash
# Round 6: s_setprio around the MFMA issue (reserved[1] = 14) / around the exact filter (15) against the default (no priorities); lab-bench library, one box
run() {
  echo "== $*"
  timeout 300 python bench.py --no-extra --no-variants --no-exchange-probe --regions 1 --cpu-seconds 2 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('ms_per_step %.4f score %.4f solve %.4f clock %4.0f kcycles %.1f parity %s' % (d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], r['avg_launch_ms']*r['shader_clock_mhz'], d['result'].get('parity_vs_oracle')))
"
}
for rep in 1 2; do
for r in "0 14 0 0" "0 15 0 0" "0 0 0 7"; do
  run --reserved $r
  run --serial --reserved $r
  run --hyps 131072 --reserved $r
  run --config c3 --reserved $r
done
done
