#!/bin/bash
# the multivariate per-SNP stage: the fixed kernel of (d = 3, c = 2) against the run-time kernel on the same shape, and six traits
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_15; mkdir -p $OUT
{
echo "== fixed kernel, d = 3";            timeout 200 python scripts/mvlmm_probe.py 10000 8192 3 1
echo "== run-time kernel, d = 3 (GEMMA_HIP_MVLMM_RT=1)"; GEMMA_HIP_MVLMM_RT=1 timeout 200 python scripts/mvlmm_probe.py 10000 8192 3 1
echo "== run-time kernel, d = 6";         timeout 300 python scripts/mvlmm_probe.py 10000 4096 6 1
} > $OUT/mvlmm_rt_probe.txt 2>&1
grep -v amdgpu.ids $OUT/mvlmm_rt_probe.txt
