#!/bin/bash
# round 4, fifth GPU call: structured divide & conquer merges A/B on a spectrum that does not deflate, then the full GPU suite and the bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_5; mkdir -p $OUT
run() { echo "== $*" >> $OUT/eigh.txt; env "$@" GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N $KIND >> $OUT/eigh.txt 2>&1; }
KIND=kin
N=2000; run GEMMA_HIP_EIGH_STAGES=1
N=4096; run A=default
N=8192; run A=default
N=8192; run GEMMA_HIP_EIGH_DC_STRUCT=0
if ! grep -q "eigh n=8192 (kin)" $OUT/eigh.txt; then echo "eigensolver failed: stopping"; tail -20 $OUT/eigh.txt; exit 3; fi
N=20000; run A=default
N=20000; run GEMMA_HIP_EIGH_DC_STRUCT=0
KIND=""
N=20000; run A=default
grep -E "==|eigh|dense|tridiag" $OUT/eigh.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
timeout 1200 python bench.py > $OUT/bench.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_5/bench.jsonl').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['stage_ms_per_step'])
cb=l['cpu_baseline']; print({k:cb[k] for k in cb if k.startswith('gpu_vs')})
print(l['config']['setup'].get('eigen_s'), l['config']['setup'].get('eigen_stages_s'), l['roofline']['frac'])
c=l.get('c4_leg',{}); print(c.get('value'), c.get('setup'), c.get('seconds'))
PY
