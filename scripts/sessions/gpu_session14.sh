#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 512 > gpurun_out/bench_full.log 2>&1
timeout 300 python bench.py --individuals 5000 --a-mode 4 --steps 3 --warmup 1 --cpu-sample 256 > gpurun_out/bench_c2.log 2>&1
cat gpurun_out/pytest_gpu.log
for f in bench_full bench_c2; do tail -1 gpurun_out/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step'], d.get('cpu_baseline',{}).get('gpu_vs_oracle_max_rel_err'))"; done
