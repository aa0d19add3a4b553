#!/bin/bash
# per-kernel split of the final eigensolver at n = 20000 on a kinship-like spectrum (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_20; mkdir -p $OUT
GEMMA_HIP_EIGH_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o e -- python scripts/eigh_probe.py 20000 kin > $OUT/eigh_profiled.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eigh_kernel_stats.csv \;
rm -rf $OUT/prof
grep -E "eigh" $OUT/eigh_profiled.txt
head -25 $OUT/eigh_kernel_stats.csv | cut -c1-150
