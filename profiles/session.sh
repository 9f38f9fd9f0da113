#!/bin/sh
# One gpurun call: GPU test suite + the bench presets + the tile-grid A/B.  sh profiles/session.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -5 $O/${TAG}_pytest.log
python3 bench.py --steps 100 --warmup 20 > $O/${TAG}_bench_headline.json 2> $O/${TAG}_bench_headline.err
python3 bench.py --config c3 --steps 200 --warmup 20 --no-cpu --no-variants > $O/${TAG}_bench_c3.json 2>> $O/${TAG}_bench_headline.err
python3 bench.py --config c4 --steps 30 --warmup 5 --no-cpu --no-variants > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench_headline.err
for cfg in c3 c4; do
  for mb in 4 8 16 32; do
    python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu --no-variants --reserved 0 0 $mb > $O/${TAG}_ab_${cfg}_grid2d_mb$mb.json 2>/dev/null
  done
  python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu --no-variants --reserved 0 1 > $O/${TAG}_ab_${cfg}_tileloop.json 2>/dev/null
done
python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/${TAG}_gpus2.out 2> $O/${TAG}_gpus2.err; echo "rc=$?" >> $O/${TAG}_gpus2.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/${TAG}_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(os.path.basename(f), "ms/step %.4f"%d["ms_per_step"], "score ms %.4f"%r["avg_launch_ms"], "frac %.4f"%r["frac"], "clk %.0f"%r["shader_clock_mhz"], "grid", d["config"]["kernel"]["grid"], d["result"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/${TAG}_gpus2.err
