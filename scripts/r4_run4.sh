#!/bin/bash
# round 4, fourth GPU call: K-sliced skinny product A/B, eigensolver tests, counter passes on the timed kernels, default bench with c4_leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_4; mkdir -p $OUT
run() { echo "== $*" >> $OUT/eigh.txt; env "$@" GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N >> $OUT/eigh.txt 2>&1; }
N=1000; run GEMMA_HIP_EIGH_STAGES=2
N=1500; run GEMMA_HIP_EIGH_STAGES=2
N=20000
run A=default
if ! grep -q "eigh n=20000" $OUT/eigh.txt; then echo "eigensolver failed at n = 20000: stopping"; cat $OUT/eigh.txt | tail -20; exit 3; fi
run GEMMA_HIP_EIGH_KSLICES=1
run GEMMA_HIP_EIGH_KSLICES=8
run A=default
N=8192; run GEMMA_HIP_EIGH_STAGES=2
grep -E "==|eigh|dense" $OUT/eigh.txt
timeout 900 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_two_rank.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -4 $OUT/pytest_gpu.txt
bash scripts/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1
tail -40 $OUT/pmc.log | grep -E "sparse2_kernel|pass" | head -30
timeout 1200 python bench.py > $OUT/bench.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_4/bench.jsonl').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['stage_ms_per_step'])
cb=l['cpu_baseline']; print({k:cb[k] for k in cb if k.startswith('gpu_vs')})
print(l['config']['setup'].get('eigen_s'), l['config']['setup'].get('eigen_stages_s'), l['roofline']['frac'])
print(json.dumps(l.get('c4_leg'))[:1800])
PY
