#!/bin/bash
mkdir -p gpurun_out
{
for ab in 0 1 2 3 7; do
GEMMA_HIP_UTX_I8=1 GEMMA_HIP_I8_ABLATE=$ab python scripts/i8_probe.py
done
for gm in 2 4 16; do
GEMMA_HIP_UTX_I8=1 GEMMA_HIP_I8_GM=$gm python scripts/i8_probe.py
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s35_probe.log
cat gpurun_out/s35_probe.log
