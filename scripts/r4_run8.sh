#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_8; mkdir -p $OUT
timeout 900 python scripts/eigh_sweep.py > $OUT/sweep.txt 2>&1; echo "sweep rc=$?" >> $OUT/sweep.txt
cat $OUT/sweep.txt | grep -v amdgpu.ids
for pn in launch persist; do
  echo "== n=50000 GEMMA_HIP_EIGH_PANEL=$pn" >> $OUT/eigh50k.txt
  GEMMA_HIP_EIGH_PANEL=$pn GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py 50000 kin >> $OUT/eigh50k.txt 2>&1
done
grep -E "==|eigh" $OUT/eigh50k.txt
