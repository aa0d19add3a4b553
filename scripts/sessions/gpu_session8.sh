#!/bin/bash
mkdir -p gpurun_out
for w in 4 8; do for a in 0 1 2 3 4 5 7; do GEMMA_HIP_GEMM_WAVES=$w GEMMA_HIP_GEMM_ABLATE=$a timeout 120 python scripts/gemm_probe.py 2>&1 | tail -1; done; done | tee gpurun_out/gemm_ablate.log
