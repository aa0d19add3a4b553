#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  GEMMA_HIP_GEMM_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma" -d $GRAFT_REPO_ROOT/gpurun_out/pmcg_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > $GRAFT_REPO_ROOT/gpurun_out/pmcg_$tag.log 2>&1
  echo "pmcg $tag exit $?"
done
cd $GRAFT_REPO_ROOT
# 2-rank path on one GPU (gloo carries the collectives): exercises broadcast / barrier / MAX-reduce of bench.py
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --individuals 4096 --batch 4096 --kin-snps 4096 --steps 2 --warmup 1 > gpurun_out/bench_2rank.log 2>&1
echo "2-rank exit $?"; tail -1 gpurun_out/bench_2rank.log | cut -c1-700
