#!/bin/bash
mkdir -p gpurun_out
{
for ab in 0 1 16 17 2 19; do
GEMMA_HIP_GEMM_PIPE=1 GEMMA_HIP_GEMM_ABLATE=$ab python scripts/gemm_probe.py
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s24_probe.log
cat gpurun_out/s24_probe.log
