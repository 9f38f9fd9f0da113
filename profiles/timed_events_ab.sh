for rep in 1 2; do
for ev in on off; do
for h in "" "--hyps 131072"; do
python3 bench.py --no-cpu --no-variants --no-extra --timed-events $ev $h 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('events $ev $h', round(d['ms_per_step'],4), '%.4g' % d['value'], d['roofline']['frac'])"
done; done; done
