#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x -k "symmetric or 4096" 2>&1 | tail -3
{
GEMMA_HIP_EIGH_TIMING=1 python scripts/eigh_probe.py 20000
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_SYMV_MIN=6000 python scripts/eigh_probe.py 20000
GEMMA_HIP_EIGH_TIMING=1 python scripts/eigh_probe.py 50000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s58_eigh.log
cat gpurun_out/s58_eigh.log
