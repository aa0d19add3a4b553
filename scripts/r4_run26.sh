#!/bin/bash
# the records kernel with its genotype product on v_mfma_i32_16x16x64_i8 (variant 6, prototype) against the shipped kernel (variant 3 = 8 x 1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_26; mkdir -p $OUT
B=scripts/abl_bin/kb8
{
echo "== variant 3 (shipped: i8gemm_sparse2_kernel_t<1>), raster 2"; RASTER=2 REPS=4 timeout 60 $B 20000 20000 3 0
echo "== variant 6 (genotype product on 16x16x64), raster 2";       RASTER=2 REPS=4 timeout 60 $B 20000 20000 6 0
echo "== variant 6, ragged sizes";                                  RASTER=2 REPS=1 timeout 60 $B 5003 3001 6 0
echo "== variant 6, B_MODE=1 (zero digits: the schedule's own time)"; B_MODE=1 RASTER=2 REPS=4 timeout 60 $B 20000 20000 6 0
} > $OUT/g16.txt 2>&1
grep -E "==|variant" $OUT/g16.txt
