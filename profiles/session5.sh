R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['solve_kernel_avg_ms'],4), d['config']['kernel']['grid'])"; }
for c in 16 32 64 128; do python3 bench.py --hyps 131072 --steps 100 --warmup 10 --no-cpu --no-variants --reserved 0 0 $c 2>/dev/null | show "H131072 cols$c"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02b_h131072 -o k -- python3 $R/bench.py --hyps 131072 --steps 100 --warmup 10 --no-cpu --no-variants > /dev/null 2>&1
cut -d, -f1-4,6,7 $O/r02b_h131072/k_kernel_stats.csv | sed 's/(.*)"/"/' | head -8
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/r02b_h131072/k_kernel_trace.csv")))
rows=[r for r in rows if 'sfm' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# one step in the middle
i0=len(rows)//2
while 'solve' not in rows[i0]['Kernel_Name']: i0+=1
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+6]:
    print(r['Kernel_Name'].split('(')[0][-40:], int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0)
PY
