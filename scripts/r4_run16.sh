#!/bin/bash
# end of round 4: the new chase-variant test, then the driver's bench command once more (another box: the records kernel moves with its power cap)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_16; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_eigh.py -x -q -k "hand_over_variants or default_path" > $OUT/eigh_tests.txt 2>&1; tail -3 $OUT/eigh_tests.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_16/bench_driver_cmd.jsonl').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['stage_ms_per_step'], l['roofline']['frac'], l['config']['setup']['eigen_s'], l['amdahl']['serial_fraction_at_8'])
print(l['dosage_path']['value'], l['digits7_leg']['value'], l['c4_leg']['value'], l['c4_leg']['setup']['eigen_s'], l['e2e']['wall_s'])
PY
