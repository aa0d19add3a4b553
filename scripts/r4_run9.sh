#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_9; mkdir -p $OUT
GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py 50000 kin > $OUT/eigh50k.txt 2>&1
timeout 300 python scripts/eigh_sweep.py 8000 20001 33000 40000 >> $OUT/eigh50k.txt 2>&1
grep -E "eigh|n=|worst" $OUT/eigh50k.txt
