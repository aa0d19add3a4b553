#!/bin/bash
# variant 7 (16-row forms) against the shipped kernel on EVERY plane entry, over the shapes the library meets
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_30; mkdir -p $OUT
B=scripts/abl_bin/kb10
{
FULLCMP=1 RASTER=2 REPS=1 timeout 60 $B 20000 20000 7 0
FULLCMP=1 RASTER=0 REPS=1 timeout 60 $B 20000 4096 7 0
FULLCMP=1 RASTER=2 REPS=1 DIGITS=7 timeout 60 $B 16640 4096 7 0
FULLCMP=1 RASTER=2 REPS=1 FUSE=0 timeout 60 $B 5003 3001 7 0
FULLCMP=1 RASTER=2 REPS=1 timeout 60 $B 5003 3001 7 0
FULLCMP=1 RASTER=2 REPS=1 timeout 60 $B 300 700 7 0
FULLCMP=1 RASTER=2 REPS=1 timeout 60 $B 200 257 7 0
FULLCMP=1 RASTER=1 REPS=1 timeout 60 $B 100 64 7 0
FULLCMP=1 RASTER=2 REPS=1 DIGITS=7 timeout 60 $B 100 300 7 0
FULLCMP=1 RASTER=4 REPS=1 timeout 60 $B 33000 2048 7 0
} > $OUT/fullcmp.txt 2>&1
grep -E "FULLCMP|variant|rror" $OUT/fullcmp.txt
