#!/bin/bash
# other BASELINE configs as smoke + timing: C2 (n=5000, -lmm 4) and C4-sized (n=50000, -lmm 1, smaller SNP block)
mkdir -p gpurun_out
GEMMA_HIP_EIGH_TIMING=1 timeout 600 python bench.py --individuals 5000 --batch 20000 --kin-snps 20000 --a-mode 4 --steps 3 --warmup 1 --cpu-sample 512 > gpurun_out/bench_c2.log 2>&1
echo "exit $?" >> gpurun_out/bench_c2.log
GEMMA_HIP_EIGH_TIMING=1 timeout 1500 python bench.py --individuals 50000 --batch 10000 --kin-snps 10000 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_c4.log 2>&1
echo "exit $?" >> gpurun_out/bench_c4.log
for f in bench_c2 bench_c4; do echo == $f; grep gemma_hip_eigh gpurun_out/$f.log; tail -2 gpurun_out/$f.log | cut -c1-2200; done
