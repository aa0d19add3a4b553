#!/bin/bash
# Lease 26: look-ahead with its threshold (default: trailing updates of 20 000 rows or more): n = 50 000 with the default / always / never,
# n = 20 000 and 33 000 with the default; bit identity again.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_26}; mkdir -p $OUT
timeout 600 python scripts/exp/r6_25.py 2>&1 | tail -3
for N in 20000 33000; do
  echo "== n = $N, default"
  GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N kin 2>&1 | grep -v "^$" | tail -3
  echo "== n = $N, GEMMA_HIP_EIGH_LOOKAHEAD=0"
  GEMMA_HIP_EIGH_LOOKAHEAD=0 GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N kin 2>&1 | grep -v "^$" | tail -3
done
for LA in "" 0 1 12000 30000 "" 0; do
  echo "== n = 50000, GEMMA_HIP_EIGH_LOOKAHEAD='$LA'"
  GEMMA_HIP_EIGH_LOOKAHEAD=$LA GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py 50000 kin 2>&1 | grep -v "^$" | tail -3
done
