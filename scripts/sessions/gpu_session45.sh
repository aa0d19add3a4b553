#!/bin/bash
mkdir -p gpurun_out
python bench.py --individuals 5000 --batch 20000 --kin-snps 20000 --a-mode 4 --steps 4 --cpu-sample 0 > gpurun_out/s45_bench_c2.log 2>&1
tail -1 gpurun_out/s45_bench_c2.log | cut -c1-260; tail -1 gpurun_out/s45_bench_c2.log | grep -o '"stage_ms_per_step[^}]*}'; tail -1 gpurun_out/s45_bench_c2.log | grep -o '"fp64_gemm_path": {"value": [0-9.]*'
GEMMA_HIP_EIGH_TIMING=1 python bench.py --individuals 50000 --batch 10000 --kin-snps 10000 --steps 2 --warmup 1 --cpu-sample 0 --fp64-steps 1 > gpurun_out/s45_bench_c4.log 2>&1
tail -1 gpurun_out/s45_bench_c4.log | cut -c1-260; tail -1 gpurun_out/s45_bench_c4.log | grep -o '"stage_ms_per_step[^}]*}'; tail -1 gpurun_out/s45_bench_c4.log | grep -o '"fp64_gemm_path": {"value": [0-9.]*'; tail -1 gpurun_out/s45_bench_c4.log | grep -o '"setup": {[^}]*}'
