#!/bin/bash
# both products on the 16-row forms, no lane swaps (variant 7, prototype) against the shipped kernel (variant 3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_29; mkdir -p $OUT
B=scripts/abl_bin/kb9
{
echo "== variant 3 (shipped), raster 2"; RASTER=2 REPS=4 timeout 60 $B 20000 20000 3 0
echo "== variant 7 (16x16x64 + sparse 16x16x128), raster 2"; RASTER=2 REPS=4 timeout 60 $B 20000 20000 7 0
echo "== variant 7, zero digits"; B_MODE=1 RASTER=2 REPS=4 timeout 60 $B 20000 20000 7 0
echo "== variant 7, ragged"; RASTER=2 REPS=1 timeout 60 $B 5003 3001 7 0
} > $OUT/g16s.txt 2>&1
grep -E "==|variant" $OUT/g16s.txt
