#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_11; mkdir -p $OUT
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_BC_DBG=1 timeout 120 python scripts/eigh_probe.py 20000 > $OUT/chase_ab.txt 2>&1
GEMMA_HIP_EIGH_TIMING=1 timeout 120 python scripts/eigh_probe.py 20000 kin >> $OUT/chase_ab.txt 2>&1
GEMMA_HIP_EIGH_TIMING=1 timeout 120 python scripts/eigh_probe.py 8192 >> $OUT/chase_ab.txt 2>&1
timeout 200 python scripts/eigh_sweep.py 8000 8001 9999 12346 20001 32768 >> $OUT/chase_ab.txt 2>&1
grep -E "eigh|n=|worst|chase|sweep" $OUT/chase_ab.txt
