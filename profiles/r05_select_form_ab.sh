R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; OUT=$O/r05_ab_select_form.txt; : > $OUT
run() { L="$1"; shift
  python3 bench.py --no-cpu --no-variants --no-extra "$@" 2>>$O/r05_sel_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s ms_per_step %.4f regions %s score %.4f solve %.4f clock %4.0f best %s' % ('$L', d['ms_per_step'], d.get('ms_per_step_regions',{}).get('all'), r['avg_launch_ms'], r['solve_kernel_avg_ms'], r['shader_clock_mhz'], d['result'].get('best_hypothesis')))" >> $OUT
}
for rep in 1 2 3; do
SFM_AMD_LIB_DIR=$R/_lib_old run "old  (VOP2 selects)"
run "new  (VOP3 selects in device_math)"
SFM_AMD_LIB_DIR=$R/_lib_old run "old  --serial" --serial
run "new  --serial" --serial
SFM_AMD_LIB_DIR=$R/_lib_old run "old  --hyps 131072 --serial" --hyps 131072 --serial
run "new  --hyps 131072 --serial" --hyps 131072 --serial
done
cat $OUT
