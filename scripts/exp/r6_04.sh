#!/bin/bash
# Round 6, lease 4: stage-1 back-transformation with 2 / 4 / 8 panels per block reflector (GEMMA_HIP_EIGH_Q1_GROUP) -> profiles/r06_eigh_q1_group.txt
{
for G in 2 4 8; do
  echo "== n = 20000 (kin), Q1 group $G"
  GEMMA_HIP_EIGH_Q1_GROUP=$G GEMMA_HIP_EIGH_TIMING=1 EIGH_PROBE_CHECK=1 timeout 300 python scripts/eigh_probe.py 20000 kin 2>&1 | grep -E "eigh|gemma_hip_eigh"
done
for G in 2 4 8; do
  echo "== n = 50000 (kin), Q1 group $G"
  GEMMA_HIP_EIGH_Q1_GROUP=$G GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py 50000 kin 2>&1 | grep -E "eigh|gemma_hip_eigh"
done
echo "== n = 8192 / 14081 (odd: embedded), Q1 group 4 and 8, residual + orthogonality"
for G in 4 8; do
  GEMMA_HIP_EIGH_Q1_GROUP=$G GEMMA_HIP_EIGH_TIMING=1 EIGH_PROBE_CHECK=1 timeout 300 python scripts/eigh_probe.py 8192 kin 2>&1 | grep -E "eigh n="
  GEMMA_HIP_EIGH_Q1_GROUP=$G GEMMA_HIP_EIGH_TIMING=1 EIGH_PROBE_CHECK=1 timeout 300 python scripts/eigh_probe.py 14081 kin 2>&1 | grep -E "eigh n="
done
} > $OUT/eigh_q1_group.txt 2>&1
cat $OUT/eigh_q1_group.txt | cut -c1-250
