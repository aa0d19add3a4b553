#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 ./scripts/mfma_f64_peak > gpurun_out/mfma_peak.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --eigen gemma > gpurun_out/bench_full.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_full.log
cd /tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --eigen torch --kin-snps 2000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/mfma_peak.log
tail -2 gpurun_out/bench_full.log | cut -c1-2500
ls -R gpurun_out/pmc_* | head -20
