#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "int8" -s 2>&1 | grep -v "^parity\|amdgpu.ids" | tail -8
for k in 1 0; do
GEMMA_HIP_I8_KERNEL=$k GEMMA_HIP_UTX_I8=1 timeout 600 python bench.py --cpu-sample 256 --steps 3 > gpurun_out/s34_bench_i8_k$k.log 2>&1
echo "kernel $k:"; tail -1 gpurun_out/s34_bench_i8_k$k.log | grep -o '"value": [0-9.]*' | head -1; tail -1 gpurun_out/s34_bench_i8_k$k.log | grep -o '"stage_ms_per_step[^}]*}'; tail -1 gpurun_out/s34_bench_i8_k$k.log | grep -o '"gpu_vs_oracle[^}]*}'
done
