#!/bin/bash
# how much of the records kernel's (power-limited) time depends on the VALUES of the digit operand
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_23; mkdir -p $OUT
B=scripts/abl_bin/kb7
{
for m in 0 1 2 3 4; do echo "== records kernel (variant 2, raster 2), B_MODE=$m"; B_MODE=$m RASTER=2 REPS=4 timeout 60 $B 20000 20000 2 0; done
for m in 0 1 3; do echo "== dense packed kernel (variant 5), B_MODE=$m"; B_MODE=$m REPS=3 timeout 60 $B 20000 20000 5 0; done
} > $OUT/bmode.txt 2>&1
grep -E "==|variant" $OUT/bmode.txt
