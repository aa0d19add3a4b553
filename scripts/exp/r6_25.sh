#!/bin/bash
# Lease 25: stage-1 look-ahead of the two-stage eigensolver (next panel's QR on a high-priority stream beside the second piece of the
# trailing update): eigen tests, bit identity against GEMMA_HIP_EIGH_LOOKAHEAD=0, stage times at n = 20 000 / 50 000 either way.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_25}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -x -q > $OUT/test_eigh.txt 2>&1; tail -3 $OUT/test_eigh.txt
timeout 600 python scripts/exp/r6_25.py 2>&1 | tail -5
for N in 20000 50000; do
  for LA in ${LA_LIST:-1 0 1 0}; do
    echo "== n = $N, GEMMA_HIP_EIGH_LOOKAHEAD=$LA"
    GEMMA_HIP_EIGH_LOOKAHEAD=$LA GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N kin 2>&1 | grep -v "^$" | tail -3
  done
done
