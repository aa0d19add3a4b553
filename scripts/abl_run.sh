#!/bin/bash
# Development loop of the int8 U^T x kernels: runs scripts/abl_bin/<binary> (scripts/i8_kernel_bench.hip built on the CPU box,
# possibly with experiment macros) for a list of "binary:variant:gm" specs, then optional rocprofv3 counter passes.
# usage: scripts/abl_run.sh <outdir> "<timing specs>" "<pmc specs>"
OUT=${1:-gpurun_out/abl}; TV=${2:-}; PV=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for spec in $TV; do
  IFS=: read b v gm <<< "$spec"
  for rep in 1 2; do timeout 120 scripts/abl_bin/$b 20000 20000 $v $gm 2>&1 | sed "s/^/[$spec] /" | tee -a "$OUT/timing.txt"; done
done
for spec in $PV; do
  IFS=: read b v gm <<< "$spec"
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 180 rocprofv3 --pmc $C --kernel-include-regex "i8gemm" --kernel-trace --output-format csv -d "$OUT/pmc_${b}_${v}_$i" -o p -- \
        scripts/abl_bin/$b 20000 20000 $v $gm > "$OUT/pmc_${b}_${v}_$i.log" 2>&1
    echo "pmc $spec pass $i ($C): rc=$?"
  done
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f.split("/")[-3], k, c, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY
