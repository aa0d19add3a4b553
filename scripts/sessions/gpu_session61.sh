#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/s61_bench.log 2>&1
tail -1 gpurun_out/s61_bench.log | cut -c1-200; tail -1 gpurun_out/s61_bench.log | grep -o '"stage_ms_per_step[^}]*}'; tail -1 gpurun_out/s61_bench.log | grep -o '"setup": {[^}]*}'
