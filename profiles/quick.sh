#!/bin/sh
# Quick GPU check used while iterating on the scoring kernel: the parity tests that cover it, then the headline bench
# (pipelined and serial) in one compact line each.  sh profiles/quick.sh [pytest targets]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
T=${*:-tests/test_gpu_prefilter.py tests/test_gpu_ransac.py}
python -m pytest $T -x -q > $O/quick_pytest.txt 2>&1
for mode in "" "--serial"; do
python bench.py $mode --no-cpu --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$mode', 'ms_per_step %.4f value %.4g score %.4f solve %.4f clk %.0f parity %s' % (d['ms_per_step'], d['value'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['result'].get('parity_vs_oracle', d['result'])))"
done
tail -4 $O/quick_pytest.txt
