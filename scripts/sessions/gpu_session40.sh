#!/bin/bash
mkdir -p gpurun_out
{
for v in 0 43 83 24 44 42; do
GEMMA_HIP_ASSOC_VARIANT=$v python scripts/assoc_probe.py
done
GEMMA_HIP_ASSOC_GRID=0 python scripts/assoc_probe.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s40_probe.log
cat gpurun_out/s40_probe.log
