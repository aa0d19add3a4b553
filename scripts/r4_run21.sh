#!/bin/bash
# dense byte-plane product: the new 256 x 256 x 64 kernel (variant 4) against the kernel it would replace for dosages (variant 5)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_21; mkdir -p $OUT
B=scripts/abl_bin/kb6
{
echo "== variant 5 (i8gemm_packed_kernel_t<false, true>)"; REPS=3 timeout 60 $B 20000 20000 5 0
echo "== variant 4 (i8gemm_dense2_kernel), gm 8";  REPS=3 timeout 60 $B 20000 20000 4 0
echo "== variant 4, gm 4";  REPS=3 timeout 60 $B 20000 20000 4 4
echo "== variant 4, gm 16"; REPS=3 timeout 60 $B 20000 20000 4 16
echo "== variant 4, ragged sizes"; REPS=1 timeout 60 $B 5003 3001 4 0
} > $OUT/dense2.txt 2>&1
cat $OUT/dense2.txt
