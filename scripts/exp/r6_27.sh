#!/bin/bash
# Lease 27: look-ahead on the launch-path panels only (the default): eigen tests incl. the bit-identity one, n = 20 000 / 33 000 / 50 000.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_27}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -x -q > $OUT/test_eigh.txt 2>&1; tail -3 $OUT/test_eigh.txt
for N in 20000 33000 50000 50000; do
  for LA in "" 0; do
    echo "== n = $N, GEMMA_HIP_EIGH_LOOKAHEAD='$LA'"
    GEMMA_HIP_EIGH_LOOKAHEAD=$LA GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N kin 2>&1 | grep -v "^$" | tail -3
  done
done
