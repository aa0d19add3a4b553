#!/bin/bash
# matrix-pipe occupancy and clock of the two dense byte-plane kernels (one counter pass each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_22; mkdir -p $OUT
B=scripts/abl_bin/kb6
for v in 4 5; do
  REPS=2 timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-include-regex "i8gemm" --kernel-trace --output-format csv -d $OUT/pmc_$v -o p -- $B 20000 20000 $v 4 > $OUT/pmc_$v.log 2>&1
  echo "variant $v rc=$?"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f.split("/")[-3], k, c, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
for f in sorted(glob.glob(out + "/pmc_*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "i8gemm" in r["Kernel_Name"]:
            print(f.split("/")[-3], r["Kernel_Name"][:40], "duration_ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
rm -rf $OUT/pmc_4 $OUT/pmc_5
