#!/bin/bash
# mvLMM per-SNP stage in two launches (EM + Wald at 2 waves/SIMD, Newton-Raphson over the queued SNPs): parity, then A/B
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_mvlmm.py tests/test_gpu_reference.py -m gpu -q 2>&1 | tail -6 | tee gpurun_out/mv_split_tests.log
for cfg in "0 2" "1 2" "1 1"; do
  set -- $cfg
  echo "== GEMMA_HIP_MV_SPLIT=$1 GEMMA_HIP_MV_EM_WAVES=$2"
  for mode in 1 4; do
    GEMMA_HIP_MV_SPLIT=$1 GEMMA_HIP_MV_EM_WAVES=$2 timeout 200 python scripts/mvlmm_probe.py 10000 8192 3 $mode 2>&1 | grep -v amdgpu.ids | grep "mvlmm batch\|p < 1e-3\|max rel"
  done
done 2>&1 | tee gpurun_out/mv_split_probe.log
