#!/bin/bash
mkdir -p gpurun_out
timeout 60 ./scripts/mfma_f64_pattern > gpurun_out/mfma_pattern.log 2>&1
timeout 900 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
GEMMA_HIP_GEMM_WAVES=4 GEMMA_HIP_EIGH_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_w4.log 2>&1
cat gpurun_out/mfma_pattern.log; cat gpurun_out/pytest_gpu.log
grep -E "gemma_hip_eigh" gpurun_out/bench_w4.log; tail -1 gpurun_out/bench_w4.log | cut -c1-1500
