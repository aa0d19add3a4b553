#!/bin/bash
# round 4, first GPU call: kernel harness variants (mask digits x raster), eigensolver stage timing (persistent panel vs launches),
# the GPU test suite, the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_1; mkdir -p $OUT
for ms in 0 1; do for ra in 0 2 1 4 8; do
  echo "== MASK_SKIP=$ms RASTER=$ra" >> $OUT/harness.txt
  MASK_SKIP=$ms RASTER=$ra timeout 120 scripts/abl_bin/kb4 20000 20000 3 0 >> $OUT/harness.txt 2>&1
done; done
for pn in launch persist launch persist; do
  echo "== GEMMA_HIP_EIGH_PANEL=$pn" >> $OUT/eigh.txt
  GEMMA_HIP_EIGH_PANEL=$pn GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 >> $OUT/eigh.txt 2>&1
done
GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 8192 >> $OUT/eigh.txt 2>&1
GEMMA_HIP_EIGH_STAGES=2 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 8192 >> $OUT/eigh.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.jsonl
cat $OUT/harness.txt | grep -E "==|ms|bad|check" | head -60
cat $OUT/eigh.txt | grep -E "==|eigh|dense" | head -40
