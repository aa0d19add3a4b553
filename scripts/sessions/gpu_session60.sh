#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "dgemm or eigh or kin" 2>&1 | tail -3
GEMMA_HIP_EIGH_TIMING=1 python scripts/eigh_probe.py 20000 2>&1 | grep -v amdgpu.ids | tail -2
