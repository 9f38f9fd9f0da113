R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { python bench.py "$@" --no-cpu --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-50s ms_per_step %.4f score %.4f solve %.4f grid %s' % ('$*', d['ms_per_step'], r['avg_launch_ms'], r['solve_kernel_avg_ms'], d['config']['kernel']['grid']))"; }
for h in 131072 262144; do
run --hyps $h --serial
run --hyps $h --serial --reserved 1
run --hyps $h --serial --reserved 0 0 64
run --hyps $h --serial --reserved 0 0 32
run --hyps $h
run --hyps $h --reserved 1
run --hyps $h --reserved 0 0 64
done
