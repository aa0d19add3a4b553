#!/bin/bash
mkdir -p gpurun_out
for w in 4 8; do GEMMA_HIP_GEMM_WAVES=$w timeout 120 python scripts/gemm_probe.py 2>&1 | tail -1; done | tee gpurun_out/gemm_probe.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dgemm or kinship or qc or loco" 2>&1 | tail -3
