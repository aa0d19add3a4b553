#!/bin/bash
# six / seven phenotypes on the new fixed kernels (two / one wavefront per workgroup) against the run-time kernel; the multivariate tests
timeout 1200 python -m pytest tests/test_gpu_mvlmm.py tests/test_gpu_workflow_files.py -m gpu -x -q -k "mvlmm or beyond or run_time or wide or six" > $OUT/mv_tests.txt 2>&1; tail -4 $OUT/mv_tests.txt
{
echo "== d = 6, fixed kernel (default)";            timeout 300 python scripts/mvlmm_probe.py 10000 4096 6 1
echo "== d = 6, run-time kernel (GEMMA_HIP_MVLMM_RT=1)"; GEMMA_HIP_MVLMM_RT=1 timeout 300 python scripts/mvlmm_probe.py 10000 2048 6 1
echo "== d = 7, fixed kernel";                      timeout 300 python scripts/mvlmm_probe.py 10000 4096 7 1
echo "== d = 3, fixed kernel (four wavefronts per workgroup, for scale)"; timeout 300 python scripts/mvlmm_probe.py 10000 8192 3 1
} > $OUT/mvlmm_d67_probe.txt 2>&1
grep -vE "amdgpu.ids" $OUT/mvlmm_d67_probe.txt | grep -E "==|mvlmm batch|null block|oracle|max"
