#!/bin/bash
# lease 13: what does q2_apply_kernel wait for?  Ablations (results wrong, timing only): 1 = no window loads / stores while the window slides, 2 = every group reads the first pack (L2-hot)
{
for A in 0 1 2; do for n in 20000 50000; do
  echo "== n = $n (kin), GEMMA_HIP_EIGH_Q2_ABLATE=$A"
  GEMMA_HIP_EIGH_Q2_ABLATE=$A GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $n kin 2>&1 | grep -E "gemma_hip_eigh n=.*two-stage"
done; done
} > $OUT/eigh_q2_ablate.txt 2>&1; cat $OUT/eigh_q2_ablate.txt | cut -c1-250
