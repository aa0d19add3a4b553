#!/bin/bash
# dense byte-plane product, 256 x 256 x 64 tiles with 8 wavefronts of 64 x 128 (variants 15 raw / 16 masked) against the shipped 128 x 256 x 128 form (10 / 11)
B=scripts/abl_bin/kb18
{
echo "== dosage-like left factor, full-range digits"
for v in 10 15 10 15; do echo "-- variant $v"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 $v 0; done
echo "== zero digits"
for v in 10 15; do echo "-- variant $v"; A_MODE=1 B_MODE=1 REPS=3 timeout 60 $B 20000 20000 $v 0; done
echo "== gm sweep, variant 15"
for gm in 4 16; do echo "-- gm $gm"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 15 $gm; done
echo "== genotype bytes, one plane (kinship-like)"
for v in 11 16; do echo "-- variant $v"; DIGITS=1 FUSE=0 B_MODE=2 REPS=5 timeout 60 $B 20000 20000 $v 0; done
echo "== FULLCMP"
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 20000 20000 15 0
FULLCMP=1 A_MODE=1 REPS=1 DIGITS=7 FUSE=0 timeout 60 $B 5003 3001 15 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 300 700 15 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 100 130 15 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 257 129 15 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 500 257 15 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 1000 300 15 0
FULLCMP=1 REPS=1 DIGITS=1 FUSE=0 B_MODE=2 timeout 60 $B 20000 20000 16 0
FULLCMP=1 REPS=1 DIGITS=2 FUSE=0 B_MODE=2 timeout 60 $B 5003 3001 16 0
FULLCMP=1 REPS=1 DIGITS=1 FUSE=0 B_MODE=2 timeout 60 $B 200 300 16 0
} > $OUT/dense16b.txt 2>&1
cat $OUT/dense16b.txt | grep -vE "^sparse2_meta"
