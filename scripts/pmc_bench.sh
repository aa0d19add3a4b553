#!/bin/bash
# HBM-side counters of the timed kernels of bench.py, one rocprofv3 --pmc pass per counter group (MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE do not fit one pass), restricted to the kernels of the timed region by name so that the
# eigensolver's ~80 000 setup launches are not instrumented.  Usage (GPU box): scripts/pmc_bench.sh <outdir> [bench args]
OUT=${1:-gpurun_out/pmc_r02}; shift
ARGS=${@:---steps 2 --warmup 1 --cpu-sample 0 --fp64-steps 0 --kin-snps 20000 --state-file /tmp/bench_state.pt}
RX='i8gemm_packed|i8_combine|table_v2|table_reduce|lmm_assoc1|cheb_scan|cheb_search|ingest_i8'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# setup once, unprofiled: rocprofv3's counter mode crashes inside the eigensolver's launch storm; the passes load the state
python bench.py $ARGS > "$OUT/setup.log" 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-include-regex "$RX" --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- \
      python bench.py $ARGS > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($C): rc=$?"
done
ls "$OUT"/pass*/ 2>/dev/null | head -30
