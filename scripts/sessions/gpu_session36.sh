#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "int8" 2>&1 | tail -3
{
GEMMA_HIP_UTX_I8=1 python scripts/i8_probe.py
GEMMA_HIP_UTX_I8=1 GEMMA_HIP_I8_ABLATE=2 python scripts/i8_probe.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s36_probe.log
cat gpurun_out/s36_probe.log
