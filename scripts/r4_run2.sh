#!/bin/bash
# round 4, second GPU call: raster variants of the records kernel (with repeats), eigensolver A/B (panel: launches vs one persistent
# launch; chase: fences vs write-through accesses), kernel stats of the solver, the GPU test suite, the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_2; mkdir -p $OUT
for rep in 1 2; do for spec in "0 8" "1 8" "2 8" "1 16" "1 4" "2 16"; do
  set -- $spec
  echo "== RASTER=$1 RASTER_PR=$2 rep $rep" >> $OUT/harness.txt
  REPS=6 RASTER=$1 RASTER_PR=$2 timeout 120 scripts/abl_bin/kb4 20000 20000 3 0 >> $OUT/harness.txt 2>&1
done; done
grep -E "==|ms per" $OUT/harness.txt
for spec in "launch 0" "persist 1" "launch 1" "persist 0" "persist 1"; do
  set -- $spec
  echo "== GEMMA_HIP_EIGH_PANEL=$1 GEMMA_HIP_EIGH_BC_SC1=$2" >> $OUT/eigh.txt
  GEMMA_HIP_EIGH_PANEL=$1 GEMMA_HIP_EIGH_BC_SC1=$2 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 >> $OUT/eigh.txt 2>&1
done
echo "== stamps sc1=1" >> $OUT/eigh.txt
GEMMA_HIP_EIGH_BC_DBG=1 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 >> $OUT/eigh.txt 2>&1
echo "== stamps sc1=0" >> $OUT/eigh.txt
GEMMA_HIP_EIGH_BC_SC1=0 GEMMA_HIP_EIGH_BC_DBG=1 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 >> $OUT/eigh.txt 2>&1
echo "== n=8192 two-stage (residual check), n=50000" >> $OUT/eigh.txt
GEMMA_HIP_EIGH_STAGES=2 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 8192 >> $OUT/eigh.txt 2>&1
GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py 50000 >> $OUT/eigh.txt 2>&1
grep -E "==|eigh|dense|chase" $OUT/eigh.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eigh -o e -- python scripts/eigh_probe.py 20000 > $OUT/prof_eigh.log 2>&1
find $OUT/prof_eigh -name "*kernel_stats.csv" -exec cp {} $OUT/eigh_kernel_stats.csv \;
find $OUT/prof_eigh -name "*kernel_trace.csv" -exec gzip -9 {} \;
head -25 $OUT/eigh_kernel_stats.csv | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_2/bench.jsonl').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['stage_ms_per_step'])
cb=l['cpu_baseline']; print({k:cb[k] for k in cb if k.startswith('gpu_vs')})
print(l.get('digits7_leg')); print(l['config']['setup'].get('eigen_s'), l['roofline']['frac'], l.get('setup_parity',{}).get('eigh_resid'))
PY
