#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -4
{
GEMMA_HIP_EIGH_TIMING=1 python scripts/eigh_probe.py 8192
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_SYMV=0 python scripts/eigh_probe.py 8192
GEMMA_HIP_EIGH_TIMING=1 python scripts/eigh_probe.py 20000
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_SYMV=0 python scripts/eigh_probe.py 20000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s54_eigh.log
cat gpurun_out/s54_eigh.log
