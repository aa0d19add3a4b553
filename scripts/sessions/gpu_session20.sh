#!/bin/bash
# round-1 evidence with the final defaults
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o r01 -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_final.log 2>&1
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  GEMMA_HIP_GEMM_SIDE_STREAM=0 timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma|lmm_assoc|ingest_lmm" -d $GRAFT_REPO_ROOT/gpurun_out/pmcf_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --kin-snps 2000 > $GRAFT_REPO_ROOT/gpurun_out/pmcf_$tag.log 2>&1
  echo "pmcf $tag exit $?"
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench_default.log | cut -c1-2600
