#!/bin/bash
mkdir -p gpurun_out
for w in 8 4; do GEMMA_HIP_GEMM_WAVES=$w timeout 120 python scripts/gemm_probe.py 2>&1 | tail -1; done | tee gpurun_out/gemm_probe.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15
