#!/bin/bash
# Lease 22: the kinship correction on bit planes (kin_i8_corr3_kernel) against the 2-bit-code kernel: parity tests, then the kernel trace of
# five 20 000-SNP blocks at n = 20 000 either way (bench.py's kinship stage, other legs off)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_22}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "kinship or kin_ or symv or tridiagonalisation" > $OUT/test_kin.txt 2>&1; tail -5 $OUT/test_kin.txt
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 1 --c4-leg 0 --e2e-snps 0 --complete-steps 0"
for C in 3 2; do
  GEMMA_HIP_KIN_CORR=$C timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$C -o b -- python bench.py --gpus 1 --steps 2 --warmup 1 $LEGS > $OUT/bench_corr$C.jsonl 2> $OUT/bench_corr$C.err
  find $OUT/prof$C -name "*kernel_stats.csv" -exec cp {} $OUT/kin_corr${C}_kernel_stats.csv \;
  rm -rf $OUT/prof$C
  echo "== GEMMA_HIP_KIN_CORR=$C"; grep -E "kin_i8|i8gemm_packed" $OUT/kin_corr${C}_kernel_stats.csv | cut -c1-150
  python - <<PY
import json
l = json.loads(open("$OUT/bench_corr$C.jsonl").read().strip().splitlines()[-1])
s = l["config"]["setup"]
print("kinship_s", s.get("kinship_s"), "roofline_kinship ms", (s.get("roofline_kinship") or {}).get("launch_ms_total"))
print("setup_parity", {k: v for k, v in (l.get("setup_parity") or {}).items() if k.startswith("kin")})
PY
done
