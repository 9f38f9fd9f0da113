R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3), d['config']['kernel'], d['result'])"; }
for cfg in c3 c4; do for k in 4 1; do python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu --no-variants --kernel $k 2>/dev/null | show "$cfg k$k"; done; done
for cfg in c3 c4; do for c in 8 16 64; do python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu --no-variants --kernel 4 --reserved 0 0 $c 2>/dev/null | show "$cfg cols$c"; done; done
for h in 16384 32768 65536 131072 262144; do for k in 4 1; do python3 bench.py --hyps $h --steps 50 --warmup 5 --no-cpu --no-variants --kernel $k 2>/dev/null | show "H$h k$k"; done; done
