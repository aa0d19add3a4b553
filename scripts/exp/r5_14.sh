#!/bin/bash
# dense byte-plane product, 128 x 128 per wavefront (variants 13 raw / 14 genotype-masked) against the 64 x 64 forms (10 / 11) and the 32-row kernel (5 / 12)
B=scripts/abl_bin/kb16
{
echo "== dosage-like left factor, full-range digits"
for v in 5 10 13 5 10 13; do echo "-- variant $v"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 $v 0; done
echo "== zero digits"
for v in 10 13; do echo "-- variant $v"; A_MODE=1 B_MODE=1 REPS=3 timeout 60 $B 20000 20000 $v 0; done
echo "== gm sweep, variant 13"
for gm in 4 16; do echo "-- gm $gm"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 13 $gm; done
echo "== genotype bytes, one plane (kinship-like)"
for v in 12 11 14; do echo "-- variant $v"; DIGITS=1 FUSE=0 B_MODE=2 REPS=5 timeout 60 $B 20000 20000 $v 0; done
echo "== FULLCMP"
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 20000 20000 13 0
FULLCMP=1 A_MODE=1 REPS=1 DIGITS=7 FUSE=0 timeout 60 $B 5003 3001 13 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 300 700 13 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 100 130 13 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 257 129 13 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 500 257 13 0
FULLCMP=1 REPS=1 DIGITS=1 FUSE=0 B_MODE=2 timeout 60 $B 20000 20000 14 0
FULLCMP=1 REPS=1 DIGITS=2 FUSE=0 B_MODE=2 timeout 60 $B 5003 3001 14 0
FULLCMP=1 REPS=1 DIGITS=1 FUSE=0 B_MODE=2 timeout 60 $B 200 300 14 0
} > $OUT/dense16w.txt 2>&1
cat $OUT/dense16w.txt | grep -vE "^sparse2_meta"
