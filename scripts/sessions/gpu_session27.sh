#!/bin/bash
mkdir -p gpurun_out
{
for ab in 0 1 2 3; do
GEMMA_HIP_GEMM_PIPE=2 GEMMA_HIP_GEMM_ABLATE=$ab python scripts/gemm_probe.py
done
GEMMA_HIP_GEMM_PIPE=2 GEMMA_HIP_GEMM_XLDS=24576 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=2 GEMMA_HIP_GEMM_XLDS=24576 GEMMA_HIP_GEMM_ABLATE=3 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=2 GEMMA_HIP_GEMM_GM=8 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=2 GEMMA_HIP_GEMM_GM=2 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=2 GEMMA_HIP_GEMM_SIDE_STREAM=0 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=2 python scripts/gemm_probe.py 19968 19968
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s27_probe.log
cat gpurun_out/s27_probe.log
