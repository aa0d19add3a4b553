#!/bin/bash
# td_col / td_w1 with the panel index dealt over four wavefronts; symmetric-SYMV segment width 512 vs 1024
timeout 600 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -4
for sg in 1024 512; do
  echo "SEG=$sg"
  GEMMA_HIP_EIGH_SEG=$sg GEMMA_HIP_EIGH_TIMING=1 timeout 200 python scripts/eigh_probe.py 20000 2>&1 | grep -v amdgpu.ids | tail -2
done
GEMMA_HIP_EIGH_SEG=512 GEMMA_HIP_EIGH_SYMV_MIN=5120 GEMMA_HIP_EIGH_TIMING=1 timeout 200 python scripts/eigh_probe.py 20000 2>&1 | grep -v amdgpu.ids | tail -2
