for b in 0 256 512 1024 1280 0; do
  echo "== ablation bits $b"
  SFM_DBG_BITS=$b timeout 300 python bench.py --no-extra --no-variants --no-cpu --no-exchange-probe --regions 1 --serial --reserved 0 0 0 7 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('serial ms_per_step %.4f score %.4f solve %.4f clock %4.0f kcycles %.1f' % (d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], r['avg_launch_ms']*r['shader_clock_mhz']))
"
done
