#!/bin/bash
# Round 6, lease 1: is the records kernel's clock set by the socket power cap?    -> profiles/r06_power_table.txt
# scripts/abl_bin/kb6 = scripts/i8_kernel_bench.hip on the shipped headers; kb6_nos / kb6_nod = the 16-row records kernel with the sparse
# (mask) / the dense (genotype) matrix instructions compiled out (timing only); mfma_power_probe6 = register-only matrix instructions.
{
echo "== rocm-smi"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -vE "^=|^$" | head -20
echo "== shipped 16-row records kernel (variant 7, raster 1), real / zero digits"
SMI=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6 20000 20000 7 0
SMI=1 B_MODE=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6 20000 20000 7 0
echo "== the same loop without the sparse (mask) instructions, real / zero digits"
SMI=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6_nos 20000 20000 7 0
SMI=1 B_MODE=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6_nos 20000 20000 7 0
echo "== the same loop without the dense (genotype) instructions, real / zero digits"
SMI=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6_nod 20000 20000 7 0
SMI=1 B_MODE=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6_nod 20000 20000 7 0
echo "== 32-row records kernel (variant 3), real digits"
SMI=1 RASTER=1 REPS=24 timeout 120 scripts/abl_bin/kb6 20000 20000 3 0
echo "== dense byte-plane kernel on the 16-row instruction (variant 10, dosage-like left operand), real / zero digits"
SMI=1 A_MODE=1 REPS=24 timeout 120 scripts/abl_bin/kb6 20000 20000 10 0
SMI=1 A_MODE=1 B_MODE=1 REPS=24 timeout 120 scripts/abl_bin/kb6 20000 20000 10 0
echo "== register-only matrix instructions (no memory traffic), genotype-like left operand"
SMI=1 timeout 300 scripts/abl_bin/mfma_power_probe6 400000
echo "== the same, left operand almost all zero (what the mask product multiplies)"
SMI=1 A_ZERO=1 timeout 300 scripts/abl_bin/mfma_power_probe6 400000
} > $OUT/power_table.txt 2>&1
grep -E "==|variant|smi|digits" $OUT/power_table.txt | cut -c1-330
