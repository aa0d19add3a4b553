#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_27; mkdir -p $OUT
B=scripts/abl_bin/kb8
{
echo "== variant 3 (shipped), raster 2"; RASTER=2 REPS=4 timeout 60 $B 20000 20000 3 0
echo "== variant 6 (16x16x64, second schedule), raster 2"; RASTER=2 REPS=4 timeout 60 $B 20000 20000 6 0
echo "== variant 6, zero digits"; B_MODE=1 RASTER=2 REPS=4 timeout 60 $B 20000 20000 6 0
echo "== variant 6, ragged"; RASTER=2 REPS=1 timeout 60 $B 5003 3001 6 0
} > $OUT/g16.txt 2>&1
grep -E "==|variant" $OUT/g16.txt
