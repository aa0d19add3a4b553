#!/bin/bash
mkdir -p gpurun_out
{
GEMMA_HIP_GEMM_PIPE=2 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=2 python scripts/gemm_probe.py 5000 20000 5
GEMMA_HIP_GEMM_PIPE=2 python scripts/gemm_probe.py 4096 4096 5
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s26_probe.log
cat gpurun_out/s26_probe.log
GEMMA_HIP_GEMM_PIPE=2 timeout 900 python -m pytest tests -m gpu -q -x -k "dgemm or kin or gemm or eigh" 2>&1 | tail -4
