#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
GEMMA_HIP_GEMM_WAVES=4 GEMMA_HIP_EIGH_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_w4.log 2>&1
GEMMA_HIP_GEMM_WAVES=8 GEMMA_HIP_EIGH_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_w8.log 2>&1
cat gpurun_out/pytest_gpu.log
for f in bench_w4 bench_w8; do echo "== $f"; grep -E "gemma_hip_eigh" gpurun_out/$f.log; tail -1 gpurun_out/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d['stage_ms_per_step'], d['config']['setup'])"; done
