#!/bin/bash
# counters of the dense byte-plane kernels (variant 5 = 32-row, 10 = 16-row): clock, matrix-pipe occupancy, LDS activity
B=scripts/abl_bin/kb19
for v in 5 10; do
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM"; do
    tag=$(echo $C | cut -c1-10)
    A_MODE=1 REPS=2 timeout 120 rocprofv3 --pmc $C --kernel-include-regex "i8gemm_packed|i8gemm_dense16" --kernel-trace --output-format csv -d $OUT/pmc_${v}_$tag -o p -- $B 20000 20000 $v 0 > $OUT/pmc_${v}_$tag.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f.split("/")[-3], k, c, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
for f in sorted(glob.glob(out + "/pmc_*/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "i8gemm" in r["Kernel_Name"]]
    print(f.split("/")[-3], "duration_ms", ["%.2f" % x for x in d])
PY
find $OUT -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} +
