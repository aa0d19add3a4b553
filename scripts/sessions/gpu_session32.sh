#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "int8 or plink" -s 2>&1 | grep -v "^parity" | tail -12
GEMMA_HIP_UTX_I8=1 timeout 600 python bench.py --cpu-sample 256 > gpurun_out/s32_bench_i8.log 2>&1
tail -1 gpurun_out/s32_bench_i8.log | cut -c1-200; tail -1 gpurun_out/s32_bench_i8.log | grep -o '"stage_ms_per_step[^}]*}'; tail -1 gpurun_out/s32_bench_i8.log | grep -o '"gpu_vs_oracle[^}]*}'; tail -3 gpurun_out/s32_bench_i8.log | head -2 | cut -c1-300
