#!/bin/bash
mkdir -p gpurun_out
{
GEMMA_HIP_GEMM_PIPE=1 GEMMA_HIP_GEMM_XLDS=24576 GEMMA_HIP_GEMM_ABLATE=19 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 GEMMA_HIP_GEMM_XLDS=24576 GEMMA_HIP_GEMM_ABLATE=0 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 GEMMA_HIP_GEMM_ABLATE=51 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 GEMMA_HIP_GEMM_ABLATE=32 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 GEMMA_HIP_GEMM_ABLATE=0 python scripts/gemm_probe.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s25_probe.log
cat gpurun_out/s25_probe.log
