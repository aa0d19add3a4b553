#!/bin/bash
# Lease 30: first pass with the QC buffers kept between blocks, kinship uploads beside the previous block's kernels: kinship / QC / file-workflow
# tests, then config 3 from files (bench.py's e2e leg alone) twice.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_30}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "kin or qc or workflow or file or loco or two_rank" > $OUT/test_kin_files.txt 2>&1; tail -4 $OUT/test_kin_files.txt
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --complete-steps 0"
for R in 1 2; do
  timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 $LEGS --e2e-snps 1000000 > $OUT/bench_e2e_$R.jsonl 2> $OUT/bench_e2e_$R.err
  python - <<PY
import json
l = json.loads(open("$OUT/bench_e2e_$R.jsonl").read().strip().splitlines()[-1])
print(json.dumps(l.get("e2e"), indent=0)[:900])
print("kinship_s", l["config"]["setup"].get("kinship_s"))
PY
done
