#!/bin/bash
# lease 19: raster sweep on the shipped 16-row records kernel (round 4's sweep was taken on the 32-row kernel): row blocks of the cross-XCD
# super-patch (RASTER = 0 per-XCD ranges, 1, 2, 4, 8) x tile rows of the per-XCD patch (RASTER_PR = 4, 8, 16)
K=scripts/abl_bin/kb6
{
for R in 0 1 2 4 8; do for PR in 4 8 16; do
  [ $R = 0 ] && [ $PR != 8 ] && continue
  echo "== RASTER=$R RASTER_PR=$PR"
  RASTER=$R RASTER_PR=$PR REPS=12 timeout 120 $K 20000 20000 7 0 | grep variant
done; done
echo "== again RASTER=1 RASTER_PR=8 (drift check)"; RASTER=1 RASTER_PR=8 REPS=12 timeout 120 $K 20000 20000 7 0 | grep variant
} > $OUT/raster_sweep.txt 2>&1; cat $OUT/raster_sweep.txt | cut -c1-160
