#!/bin/bash
# lease 10: stage-2 back-transformation with -T in the pack and two window tiles interleaved in the third product -> profiles/r06_eigh_q2_interleave.txt
timeout 1500 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_two_rank.py -m gpu -q -x > $OUT/pytest_eigh.txt 2>&1; tail -4 $OUT/pytest_eigh.txt
{
for n in 20000 50000; do
  echo "== n = $n (kin)"
  GEMMA_HIP_EIGH_TIMING=1 EIGH_PROBE_CHECK=$([ $n = 20000 ] && echo 1 || echo 0) timeout 600 python scripts/eigh_probe.py $n kin 2>&1 | grep -E "eigh|gemma_hip_eigh"
done
} > $OUT/eigh_q2.txt 2>&1; cat $OUT/eigh_q2.txt | cut -c1-250
