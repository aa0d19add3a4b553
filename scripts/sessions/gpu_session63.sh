#!/bin/bash
# panel Gram matrices from the tridiagonalisation instead of a one-workgroup GEMM per panel in the back-transformation
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -4
for f in 1 0; do
  GEMMA_HIP_EIGH_PANEL_S=$f GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 2>&1 | tail -6
done
