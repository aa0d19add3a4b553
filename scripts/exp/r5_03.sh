#!/bin/bash
# the 16-row records kernel on a CU partition beside a stand-in for the stages behind it (scripts/i8_kernel_bench.hip CU_SPLIT):
# mask bit c = CU c % 32 of XCD c / 32 (r5_02), so CU_MODE=1 with CU_SPLIT=32 takes 4 CUs out of EVERY XCD, CU_MODE=0 the whole XCD 0
B=scripts/abl_bin/kb13
{
echo "== baseline (all CUs)"; RASTER=1 REPS=4 timeout 60 $B 20000 20000 7 0
for k in 16 32 48 64; do
echo "== split $k, spread over the XCDs"; CU_SPLIT=$k CU_MODE=1 RASTER=1 REPS=4 timeout 60 $B 20000 20000 7 0
done
echo "== split 32, one whole XCD"; CU_SPLIT=32 CU_MODE=0 RASTER=1 REPS=4 timeout 60 $B 20000 20000 7 0
echo "== baseline again"; RASTER=1 REPS=4 timeout 60 $B 20000 20000 7 0
} > $OUT/cu_split_r16.txt 2>&1
grep -E "==|variant|CU_SPLIT" $OUT/cu_split_r16.txt
