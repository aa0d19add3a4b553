#!/bin/bash
# panel dots fused into the symmetric SYMV launch, batched T factors, aligned back-transform panels; sweep the switch-over
timeout 600 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -4
for m in 12288 8192 5120; do
  echo "SYMV_MIN=$m"
  GEMMA_HIP_EIGH_SYMV_MIN=$m GEMMA_HIP_EIGH_TIMING=1 timeout 200 python scripts/eigh_probe.py 20000 2>&1 | grep -v amdgpu.ids | tail -2
done
