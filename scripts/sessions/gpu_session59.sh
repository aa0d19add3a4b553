#!/bin/bash
mkdir -p gpurun_out
{
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_SYMV_MIN=2048 python scripts/eigh_probe.py 20000
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_SYMV_MIN=2048 GEMMA_HIP_EIGH_SYMV_NQ=2 python scripts/eigh_probe.py 20000
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_SYMV_NQ=2 python scripts/eigh_probe.py 50000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s59_eigh.log
cat gpurun_out/s59_eigh.log
