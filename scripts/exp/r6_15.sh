#!/bin/bash
# lease 15: the int32 planes through non-temporal stores (S2_R16_NT_STORE=1) against the shipped stores; counters of both
K0=scripts/abl_bin/kb6; K1=scripts/abl_bin/kb6_nt
{
FULLCMP=1 RASTER=1 REPS=1 timeout 120 $K1 20000 4096 7 0 | grep FULLCMP
FULLCMP=1 RASTER=1 REPS=1 timeout 120 $K1 5003 3001 7 0 | grep FULLCMP
for rep in 1 2 3; do
echo "== shipped stores [rep $rep]"; SMI=1 RASTER=1 REPS=24 timeout 120 $K0 20000 20000 7 0
echo "== non-temporal stores [rep $rep]"; SMI=1 RASTER=1 REPS=24 timeout 120 $K1 20000 20000 7 0
done
} > $OUT/nt_store.txt 2>&1
grep -E "==|variant|FULLCMP" $OUT/nt_store.txt | cut -c1-200
RASTER=1 bash scripts/abl_run.sh $OUT/pmc "" "kb6_nt:7:0" > $OUT/pmc_nt.txt 2>&1; grep -E "TCC|FETCH|WRITE|GRBM|MFMA_BUSY" $OUT/pmc_nt.txt | cut -c1-160
