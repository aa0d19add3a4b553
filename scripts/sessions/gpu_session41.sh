#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/s41_pytest.log
python bench.py > gpurun_out/s41_bench.log 2>&1
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --individuals 4096 --batch 4096 --kin-snps 4096 --steps 2 --warmup 1 > gpurun_out/s41_bench_2rank.log 2>&1
echo "2-rank exit $?"
cat gpurun_out/s41_pytest.log; tail -1 gpurun_out/s41_bench.log | cut -c1-230; tail -1 gpurun_out/s41_bench.log | grep -o '"stage_ms_per_step[^}]*}'; tail -1 gpurun_out/s41_bench.log | grep -o '"roofline": {[^}]*}'; tail -1 gpurun_out/s41_bench_2rank.log | cut -c1-300
