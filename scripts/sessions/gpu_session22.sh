#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --individuals 4096 --batch 4096 --kin-snps 4096 --steps 2 --warmup 1 > gpurun_out/bench_2rank.log 2>&1
echo "2-rank exit $?"; tail -1 gpurun_out/bench_2rank.log | cut -c1-900
