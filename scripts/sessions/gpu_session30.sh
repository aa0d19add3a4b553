#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/s30_pytest.log
python bench.py > gpurun_out/s30_bench.log 2>&1
GEMMA_HIP_ASSOC_GRID=0 python bench.py --cpu-sample 0 > gpurun_out/s30_bench_nogrid.log 2>&1
cat gpurun_out/s30_pytest.log; tail -1 gpurun_out/s30_bench.log | cut -c1-300; tail -1 gpurun_out/s30_bench.log | grep -o '"stage_ms_per_step.*' | cut -c1-700; tail -1 gpurun_out/s30_bench_nogrid.log | grep -o '"stage_ms_per_step[^}]*}'
