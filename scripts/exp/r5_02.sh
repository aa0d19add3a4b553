#!/bin/bash
# r16 records kernel: PREP spread over the last step (kb12_1) and PERSISTENT workgroups (variant 8) against the shipped form (variant 7, kb12_0);
# XCD dispatch under a CU mask
B0=scripts/abl_bin/kb12_0; B1=scripts/abl_bin/kb12_1
{
echo "== XCD dispatch under a CU mask"; timeout 60 scripts/abl_bin/xcc_mask_probe
echo "== variant 7 shipped, raster 1"; RASTER=1 REPS=4 timeout 60 $B0 20000 20000 7 0
echo "== variant 7 PREP spread, raster 1"; RASTER=1 REPS=4 timeout 60 $B1 20000 20000 7 0
echo "== variant 8 persistent, raster 1"; RASTER=1 REPS=4 timeout 60 $B0 20000 20000 8 0
echo "== variant 8 persistent + PREP spread, raster 1"; RASTER=1 REPS=4 timeout 60 $B1 20000 20000 8 0
echo "== variant 8 persistent, raster 2"; RASTER=2 REPS=4 timeout 60 $B0 20000 20000 8 0
echo "== variant 8 persistent, raster 0 (per-XCD ranges)"; RASTER=0 REPS=4 timeout 60 $B0 20000 20000 8 0
echo "== zero digits: variant 7"; B_MODE=1 RASTER=1 REPS=4 timeout 60 $B0 20000 20000 7 0
echo "== zero digits: variant 7 PREP spread"; B_MODE=1 RASTER=1 REPS=4 timeout 60 $B1 20000 20000 7 0
echo "== zero digits: variant 8"; B_MODE=1 RASTER=1 REPS=4 timeout 60 $B0 20000 20000 8 0
echo "== zero digits: variant 8 + PREP spread"; B_MODE=1 RASTER=1 REPS=4 timeout 60 $B1 20000 20000 8 0
echo "== variant 7 shipped again (drift check)"; RASTER=1 REPS=4 timeout 60 $B0 20000 20000 7 0
for BIN in $B0 $B1; do
echo "== FULLCMP variant 8 ($BIN)"
FULLCMP=1 RASTER=1 REPS=1 timeout 60 $BIN 20000 20000 8 0
FULLCMP=1 RASTER=2 REPS=1 DIGITS=7 timeout 60 $BIN 16640 4096 8 0
FULLCMP=1 RASTER=1 REPS=1 FUSE=0 timeout 60 $BIN 5003 3001 8 0
FULLCMP=1 RASTER=0 REPS=1 timeout 60 $BIN 5003 3001 8 0
FULLCMP=1 RASTER=1 REPS=1 timeout 60 $BIN 300 700 8 0
FULLCMP=1 RASTER=1 REPS=1 timeout 60 $BIN 200 257 8 0
FULLCMP=1 RASTER=1 REPS=1 DIGITS=7 timeout 60 $BIN 100 300 8 0
FULLCMP=1 RASTER=1 REPS=1 PERSIST_WGS=7 timeout 60 $BIN 1000 3000 8 0
done
echo "== FULLCMP variant 7 PREP spread"
FULLCMP=1 RASTER=1 REPS=1 timeout 60 $B1 20000 4096 7 0
FULLCMP=1 RASTER=1 REPS=1 DIGITS=7 timeout 60 $B1 5003 3001 7 0
FULLCMP=1 RASTER=1 REPS=1 timeout 60 $B1 200 257 7 0
} > $OUT/kernel_variants.txt 2>&1
grep -E "==|variant|FULLCMP|XCD|workgroups|violated|rror" $OUT/kernel_variants.txt
# L2 hit rate, traffic, clock of variant 8 against variant 7 (one pass per counter group)
for v in 7 8; do
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $C | cut -c1-8)
    RASTER=1 REPS=2 timeout 120 rocprofv3 --pmc $C --kernel-include-regex "i8gemm_sparse2_r16" --kernel-trace --output-format csv -d $OUT/pmc_${v}_$tag -o p -- $B0 20000 20000 $v 0 > $OUT/pmc_${v}_$tag.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print(f.split("/")[-3], c, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
for f in sorted(glob.glob(out + "/pmc_*/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "r16" in r["Kernel_Name"]]
    print(f.split("/")[-3], "duration_ms", ["%.2f" % x for x in d])
PY
find $OUT -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} +
