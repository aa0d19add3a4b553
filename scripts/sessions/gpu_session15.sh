#!/bin/bash
# final round-1 evidence: tests, default bench, rocprofv3 kernel stats of the same command, PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o r01 -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_final.log 2>&1
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma|lmm_assoc|ingest_lmm" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --kin-snps 2000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -1 gpurun_out/bench_default.log | cut -c1-3000
ls gpurun_out/prof_final/*
