#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "int8 or plink" 2>&1 | tail -3
{
python scripts/i8_probe.py
GEMMA_HIP_I8_FUSE=0 python scripts/i8_probe.py
python scripts/i8_probe.py
GEMMA_HIP_I8_FUSE=0 python scripts/i8_probe.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s39_probe.log
cat gpurun_out/s39_probe.log
