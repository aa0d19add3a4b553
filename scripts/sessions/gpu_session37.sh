#!/bin/bash
mkdir -p gpurun_out
{
GEMMA_HIP_UTX_I8=1 python scripts/i8_probe.py
GEMMA_HIP_UTX_I8=1 GEMMA_HIP_I8_ABLATE=8 python scripts/i8_probe.py
GEMMA_HIP_UTX_I8=1 python scripts/i8_probe.py 20000 4096
GEMMA_HIP_UTX_I8=1 python scripts/i8_probe.py 8192 8192 5
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s37_probe.log
cat gpurun_out/s37_probe.log
