#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/pytest_eigh.log
cat gpurun_out/pytest_eigh.log
GEMMA_HIP_EIGH_TIMING=1 timeout 600 python bench.py --steps 1 --warmup 1 --cpu-sample 256 2>&1 | grep -E "gemma_hip_eigh|metric" | cut -c1-1200
cd /tmp
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  GEMMA_HIP_GEMM_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma|lmm_assoc" -d $GRAFT_REPO_ROOT/gpurun_out/pmc2_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$tag.log 2>&1
  echo "pmc2 $tag exit $?"
done
