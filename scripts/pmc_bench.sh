#!/bin/bash
# HBM-side and pipe counters of the timed kernels of bench.py, one rocprofv3 --pmc pass per counter group (MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE do not fit one pass), restricted to the kernels of the timed region by name so that the
# eigensolver's ~80 000 setup launches are not instrumented.  Usage (GPU box): scripts/pmc_bench.sh <outdir> [bench args]
OUT=${1:-gpurun_out/pmc_r04}; shift
ARGS=${@:---steps 2 --warmup 1 --cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0 --complete-steps 0 --kin-snps 20000 --state-file /tmp/bench_state.pt}
RX='i8gemm_sparse2|sparse2_meta|i8gemm_packed|i8_combine|i8_surplus|table_v2|table_reduce|lmm_assoc1|cheb_scan|cheb_search|ingest_i8'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# setup once, unprofiled: rocprofv3's counter mode crashes inside the eigensolver's launch storm; the passes load the state
python bench.py $ARGS > "$OUT/setup.log" 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-include-regex "$RX" --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- \
      python bench.py $ARGS > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($C): rc=$?"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, len(v), sum(v) / len(v)))
with open(out + "/summary.csv", "w") as fo:
    fo.write("kernel,counter,launches,mean_per_launch\n")
    for r in rows:
        fo.write("%s,%s,%d,%.6g\n" % r)
        print("%s,%s,%d,%.6g" % r)
PY
