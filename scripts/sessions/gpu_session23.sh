#!/bin/bash
mkdir -p gpurun_out
{
python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_ABLATE=3 python scripts/gemm_probe.py
GEMMA_HIP_GEMM_PIPE=1 python scripts/gemm_probe.py 5000 20000 5
python scripts/gemm_probe.py 5000 20000 5
} > gpurun_out/s23_probe.log 2>&1
cat gpurun_out/s23_probe.log
GEMMA_HIP_GEMM_PIPE=1 timeout 900 python -m pytest tests -m gpu -q -x -k "dgemm or kin or gemm or eigh" 2>&1 | tail -4
