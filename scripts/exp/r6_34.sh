#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_34}; mkdir -p $OUT
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0"
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 $LEGS > $OUT/b.jsonl 2> $OUT/b.err; echo rc=$?
python - <<PY
import json
l = json.loads(open("$OUT/b.jsonl").read().strip().splitlines()[-1])
print(l["value"], json.dumps(l["complete_leg"]))
PY
