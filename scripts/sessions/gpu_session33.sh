#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
GEMMA_HIP_UTX_I8=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s33_prof -o i8 -- python $R/bench.py --cpu-sample 0 --steps 3 > $R/gpurun_out/s33_rocprof.log 2>&1
cd $R
grep -i "i8gemm\|i8_combine\|ingest_i8\|lmm_assoc\|grid_table\|u_digits\|u_colmax" gpurun_out/s33_prof/i8_kernel_stats.csv | cut -c1-220
