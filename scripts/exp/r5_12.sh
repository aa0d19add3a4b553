#!/bin/bash
# is the 16-row records kernel's loop bound by the LDS pipe?  variant 9 = the prototype header's kernel with digit-fragment reads compiled out
{
echo "== shipped (variant 7), real / zero digits"; RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb14 20000 20000 7 0; B_MODE=1 RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb14 20000 20000 7 0
echo "== 12 of 20 LDS reads (ABLATE=1), real / zero digits"; RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb15_a1 20000 20000 9 0; B_MODE=1 RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb15_a1 20000 20000 9 0
echo "== 6 of 20 LDS reads (ABLATE=2), real / zero digits"; RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb15_a2 20000 20000 9 0; B_MODE=1 RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb15_a2 20000 20000 9 0
} > $OUT/lds_ablation.txt 2>&1
grep -E "==|variant" $OUT/lds_ablation.txt
