#!/bin/bash
# first GPU session: parity tests, smoke, small + full bench
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 300 python bench.py --individuals 4096 --batch 4096 --kin-snps 4096 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/bench_small.log 2>&1
echo "bench small exit $?" >> gpurun_out/bench_small.log
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/bench_full.log 2>&1
echo "bench full exit $?" >> gpurun_out/bench_full.log
tail -3 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench_small.log gpurun_out/bench_full.log
