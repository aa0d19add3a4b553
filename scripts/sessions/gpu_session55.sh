#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s55_prof -o e -- python $R/scripts/eigh_probe.py 8192 > $R/gpurun_out/s55.log 2>&1
cd $R
grep "td_" gpurun_out/s55_prof/e_kernel_stats.csv | cut -c1-60,200-330 | head; grep "td_" gpurun_out/s55_prof/e_kernel_stats.csv | awk -F'",' '{print $2}' | head
