#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_17; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_eigh.py -x -q -k "hand_over_variants or default_path or two_stage_end_to_end" > $OUT/eigh_tests.txt 2>&1; tail -3 $OUT/eigh_tests.txt
GEMMA_HIP_EIGH_TIMING=1 GEMMA_HIP_EIGH_BC_DBG=1 timeout 120 python scripts/eigh_probe.py 20000 > $OUT/probe.txt 2>&1; grep -E "eigh|chase" $OUT/probe.txt
timeout 200 python scripts/eigh_sweep.py 8001 12346 20001 32768 >> $OUT/probe.txt 2>&1; grep -E "n=|worst" $OUT/probe.txt
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $OUT/bench_plain.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_17/bench_plain.jsonl').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['stage_ms_per_step'], l['roofline']['frac'])
print(l['config']['setup'].get('kinship_s'), l['config']['setup'].get('eigen_s'), l['config']['setup'].get('eigen_stages_s'), l['amdahl']['serial_fraction_at_8'])
PY
