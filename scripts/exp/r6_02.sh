#!/bin/bash
# Round 6, lease 2: five LDS stages (LDS-DMA four K-tiles ahead, 128 KiB in flight per CU) against the shipped four -> profiles/r06_i8_r16_five_stages.txt
K4=scripts/abl_bin/kb6; K5=scripts/abl_bin/kb6_st5
{
echo "== exactness of the five-stage form: every plane entry against the 32-row kernel (variant 3), K loops of 1..6 tiles and the bench shape"
for shape in "100 64" "200 257" "300 700" "500 513" "640 256" "700 1000" "5003 3001" "20000 4096"; do
  set -- $shape
  FULLCMP=1 RASTER=1 REPS=1 timeout 60 $K5 $1 $2 7 0 | grep FULLCMP
done
FULLCMP=1 RASTER=1 REPS=1 DIGITS=7 timeout 60 $K5 16640 4096 7 0 | grep FULLCMP
FULLCMP=1 RASTER=2 REPS=1 FUSE=0 timeout 60 $K5 5003 3001 7 0 | grep FULLCMP
FULLCMP=1 RASTER=1 REPS=1 FUSE=0 timeout 60 $K5 33000 2048 7 0 | grep FULLCMP
FULLCMP=1 RASTER=1 REPS=1 timeout 120 $K5 20000 20000 7 0 | grep FULLCMP
for rep in 1 2; do
echo "== four stages (shipped), real / zero digits  [rep $rep]"
SMI=1 RASTER=1 REPS=24 timeout 120 $K4 20000 20000 7 0
SMI=1 B_MODE=1 RASTER=1 REPS=24 timeout 120 $K4 20000 20000 7 0
echo "== five stages, real / zero digits  [rep $rep]"
SMI=1 RASTER=1 REPS=24 timeout 120 $K5 20000 20000 7 0
SMI=1 B_MODE=1 RASTER=1 REPS=24 timeout 120 $K5 20000 20000 7 0
done
echo "== five stages, raster 2"
SMI=1 RASTER=2 REPS=24 timeout 120 $K5 20000 20000 7 0
echo "== data side alone (dense instructions compiled out): four / five stages, zero digits"
B_MODE=1 RASTER=1 REPS=12 timeout 120 scripts/abl_bin/kb6_nod 20000 20000 7 0
B_MODE=1 RASTER=1 REPS=12 timeout 120 scripts/abl_bin/kb6_st5_nod 20000 20000 7 0
echo "== config 4's shape (n = 50000, unfused planes, B = 4096): four / five stages"
FUSE=0 RASTER=1 REPS=3 timeout 200 $K4 50000 4096 7 0
FUSE=0 RASTER=1 REPS=3 timeout 200 $K5 50000 4096 7 0
} > $OUT/five_stages.txt 2>&1
grep -E "==|variant|FULLCMP|smi" $OUT/five_stages.txt | cut -c1-260
RASTER=1 bash scripts/abl_run.sh $OUT/pmc "" "kb6:7:0 kb6_st5:7:0" > $OUT/pmc_five_stages.txt 2>&1
grep -E "mean=" $OUT/pmc_five_stages.txt | cut -c1-200
