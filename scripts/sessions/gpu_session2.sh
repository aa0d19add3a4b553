#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_full.log 2>&1
echo "bench full exit $?" >> gpurun_out/bench_full.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_r01 | head -30
for f in pytest_gpu.log bench_full.log rocprof_bench.log; do echo "== $f"; tail -n 4 gpurun_out/$f | cut -c1-1500; done
