#!/bin/bash
mkdir -p gpurun_out
{
python scripts/i8_probe.py
GEMMA_HIP_I8_GRAY=1 python scripts/i8_probe.py
python scripts/i8_probe.py
GEMMA_HIP_I8_GRAY=1 python scripts/i8_probe.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s47_probe.log
cat gpurun_out/s47_probe.log
