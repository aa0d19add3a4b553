#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_12; mkdir -p $OUT
B=scripts/abl_bin/kb5
{
echo "== baseline (all CUs, default stream)"; RASTER=2 REPS=4 timeout 100 $B 20000 20000 2 0
echo "== split 16, low bits";  CU_SPLIT=16 CU_MODE=0 RASTER=2 REPS=4 timeout 100 $B 20000 20000 2 0
echo "== split 16, spread";    CU_SPLIT=16 CU_MODE=1 RASTER=2 REPS=4 timeout 100 $B 20000 20000 2 0
echo "== split 8, spread";     CU_SPLIT=8 CU_MODE=1 RASTER=2 REPS=4 timeout 100 $B 20000 20000 2 0
echo "== split 32, spread";    CU_SPLIT=32 CU_MODE=1 RASTER=2 REPS=4 timeout 100 $B 20000 20000 2 0
echo "== split 32, low bits";  CU_SPLIT=32 CU_MODE=0 RASTER=2 REPS=4 timeout 100 $B 20000 20000 2 0
} > $OUT/cu_split.txt 2>&1
cat $OUT/cu_split.txt
