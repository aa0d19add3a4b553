#!/bin/bash
mkdir -p gpurun_out
GEMMA_HIP_EIGH_TIMING=1 python scripts/eigh_probe.py 50000 2>&1 | grep -v amdgpu.ids | tail -2
