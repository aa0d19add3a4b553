#!/bin/bash
# the mirror of the trailing matrix inside the rank-256 update (GemmArgs::mirror) against the separate pass
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -x -q > $OUT/eigh_tests.txt 2>&1; tail -3 $OUT/eigh_tests.txt
for M in fused pass; do
  echo "== GEMMA_HIP_EIGH_MIRROR=$M"
  GEMMA_HIP_EIGH_MIRROR=$M EIGH_PROBE_CHECK=1 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 kin 2>&1 | grep -E "eigh"
  GEMMA_HIP_EIGH_MIRROR=$M EIGH_PROBE_CHECK=1 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 50000 kin 2>&1 | grep -E "eigh"
done > $OUT/mirror.txt 2>&1
cat $OUT/mirror.txt
