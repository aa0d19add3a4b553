#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 ./scripts/mfma_f64_peak > gpurun_out/mfma_peak.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "parity\[n=.*c=(5|7|11|16)|parity\[BXD-generic|passed|failed|FAILED|Error|error" | cut -c1-700 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
cd /tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma|lmm_assoc|ingest_lmm" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --eigen gemma --kin-snps 2000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/mfma_peak.log
tail -12 gpurun_out/pytest_gpu.log
for t in SQ_VALU_MFMA_BUSY_CYCLES FETCH_SIZE WRITE_SIZE; do tail -2 gpurun_out/pmc_$t.log | cut -c1-300; done
find gpurun_out -name "*.csv" | head
