#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "medium or synthetic or bxd_golden or fixed_lambda or degenerate" 2>&1 | tail -4
{
python scripts/assoc_probe.py
GEMMA_HIP_ASSOC_BLOCK=4 python scripts/assoc_probe.py
GEMMA_HIP_ASSOC_BLOCK=0 python scripts/assoc_probe.py
python scripts/assoc_probe.py 8192 20000
GEMMA_HIP_ASSOC_BLOCK=0 python scripts/assoc_probe.py 8192 20000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s44_probe.log
cat gpurun_out/s44_probe.log
