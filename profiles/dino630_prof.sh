#!/bin/bash
# kernel statistics of the 630-pair dino run (profiles/dino630_probe.py); run on the GPU box through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/dino630; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o d -- python3 $R/profiles/dino630_probe.py > $O/run.txt 2>&1
cd $R
for f in $(find gpurun_out/dino630 -name "*kernel_stats.csv"); do cut -c1-200 $f | head -24; done
tail -3 $O/run.txt
