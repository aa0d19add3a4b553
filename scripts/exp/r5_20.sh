#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_mvlmm.py -m gpu -x -q > $OUT/mv_tests.txt 2>&1; tail -3 $OUT/mv_tests.txt
{
echo "== d = 8, one covariate: fixed kernel (default)"; timeout 300 python scripts/mvlmm_probe.py 10000 4096 8 1
echo "== d = 8, run-time kernel (GEMMA_HIP_MVLMM_RT=1)"; GEMMA_HIP_MVLMM_RT=1 timeout 300 python scripts/mvlmm_probe.py 10000 1024 8 1
} > $OUT/mvlmm_d8_probe.txt 2>&1
grep -E "==|mvlmm batch|null block|oracle|max|rror" $OUT/mvlmm_d8_probe.txt
