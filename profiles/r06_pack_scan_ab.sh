for r in "0 0 0 0" "0 0 0 6" "0 0 0 5" "0 0 0 0" "0 0 0 6" "0 0 0 5"; do
  echo "== reserved $r"
  timeout 300 python bench.py --no-extra --no-variants --cpu-seconds 3 --reserved $r 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('ms_per_step',d['ms_per_step'],'score',r['avg_launch_ms'],'solve',r['solve_kernel_avg_ms'],'clock',r['shader_clock_mhz'],'parity',d['result'].get('parity_vs_oracle'), 'cycles', r['avg_launch_ms']*r['shader_clock_mhz']*1e3)
"
done
