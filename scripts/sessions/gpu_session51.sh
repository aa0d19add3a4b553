#!/bin/bash
# final evidence of the round: full GPU suite, smoke, default bench, rocprofv3 kernel stats of the bench, PMC on the int8 GEMM
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/s51_pytest.log
python __graft_entry__.py smoke 2>&1 | tail -1 >> gpurun_out/s51_pytest.log
python bench.py > gpurun_out/s51_bench.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s51_prof -o r01 -- python $R/bench.py --cpu-sample 0 --fp64-steps 1 > $R/gpurun_out/s51_rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/s51_rocprof.log
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "i8gemm" -d $R/gpurun_out/s51_pmc_$tag -o pmc -- python $R/scripts/i8_probe.py 20000 20000 2 > $R/gpurun_out/s51_pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done
cd $R
cat gpurun_out/s51_pytest.log; tail -1 gpurun_out/s51_bench.log | cut -c1-2800; tail -2 gpurun_out/s51_rocprof.log | cut -c1-200
