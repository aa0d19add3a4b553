#!/bin/bash
# Lease 35: look-ahead for the one-launch panel kernel with CUs of its own (96 KiB of LDS padding keeps the update's workgroups off its CUs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for N in 20000 33000 50000; do
  for LA in 3 "" 3 ""; do
    echo "== n = $N, GEMMA_HIP_EIGH_LOOKAHEAD='$LA'"
    GEMMA_HIP_EIGH_LOOKAHEAD=$LA GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N kin 2>&1 | grep -v "^$" | grep "two-stage\|eigh n" | cut -c1-120
  done
done
