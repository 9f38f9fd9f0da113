#!/bin/bash
# kernel statistics of the configs[4] front end alone (profiles/views_front_probe.py); run on the GPU box through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/viewsfront; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NV=${NV:-36} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o d -- python3 $R/profiles/views_front_probe.py > $O/run.txt 2>&1
cd $R
for f in $(find gpurun_out/viewsfront -name "*kernel_stats.csv"); do cut -c1-150 $f | head -16; done
tail -2 $O/run.txt
