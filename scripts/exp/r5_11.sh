#!/bin/bash
# the dense byte-plane product on v_mfma_i32_16x16x64_i8 (variants 10 raw / 11 genotype-masked) against the 32-row kernel (5 / 12)
B=scripts/abl_bin/kb14
{
echo "== dosage-like left factor (bytes in [-100, 100]), digits uniform in [-128, 127]"
echo "-- variant 5 (32-row)"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 5 0
echo "-- variant 10 (16-row)"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 10 0
echo "-- variant 5 again"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 5 0
echo "-- variant 10 again"; A_MODE=1 REPS=3 timeout 60 $B 20000 20000 10 0
echo "== zero digits (the schedule's own time)"
echo "-- variant 5"; A_MODE=1 B_MODE=1 REPS=3 timeout 60 $B 20000 20000 5 0
echo "-- variant 10"; A_MODE=1 B_MODE=1 REPS=3 timeout 60 $B 20000 20000 10 0
echo "== genotype bytes on both sides (kinship-like: digits in [0, 15] as a stand-in), one plane"
echo "-- variant 12 (32-row, masked)"; DIGITS=1 FUSE=0 B_MODE=2 REPS=5 timeout 60 $B 20000 20000 12 0
echo "-- variant 11 (16-row, masked)"; DIGITS=1 FUSE=0 B_MODE=2 REPS=5 timeout 60 $B 20000 20000 11 0
echo "== FULLCMP"
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 20000 20000 10 0
FULLCMP=1 A_MODE=1 REPS=1 DIGITS=7 FUSE=0 timeout 60 $B 5003 3001 10 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 300 700 10 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 100 130 10 0
FULLCMP=1 A_MODE=1 REPS=1 timeout 60 $B 257 129 10 0
FULLCMP=1 REPS=1 DIGITS=1 FUSE=0 B_MODE=2 timeout 60 $B 20000 20000 11 0
FULLCMP=1 REPS=1 DIGITS=2 FUSE=0 B_MODE=2 timeout 60 $B 5003 3001 11 0
FULLCMP=1 REPS=1 DIGITS=1 FUSE=0 B_MODE=2 timeout 60 $B 200 300 11 0
} > $OUT/dense16.txt 2>&1
cat $OUT/dense16.txt | grep -vE "^sparse2_meta"
