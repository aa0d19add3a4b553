#!/bin/bash
# round-1 closing evidence: full GPU suite (oracle parity + reference-output parity), smoke(), default bench (cpu_baseline =
# the reference's own LMM::Analyze), rocprofv3 kernel stats of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -25 ) > gpurun_out/final_pytest.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/final_smoke.log
( time timeout 400 python bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/final_bench.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 > $R/gpurun_out/final_prof.log 2>&1
cd $R
find gpurun_out/final_prof -name "*kernel_trace.csv" -delete
tail -12 gpurun_out/final_pytest.log; cat gpurun_out/final_smoke.log; tail -5 gpurun_out/final_bench.log | cut -c1-3000
f=$(find gpurun_out/final_prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-220
