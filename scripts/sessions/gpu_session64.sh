#!/bin/bash
# where the tridiagonalisation's 4.4 s go: per-kernel totals vs wall
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/eig_prof -o eig -- python $R/scripts/eigh_probe.py 20000 > $R/gpurun_out/eig_prof.log 2>&1
tail -3 $R/gpurun_out/eig_prof.log
f=$(ls $R/gpurun_out/eig_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $R/gpurun_out/eig_prof/*kernel_stats.csv | head -1)
head -25 $f | cut -c1-200
find $R/gpurun_out/eig_prof -name "*kernel_trace.csv" -delete
