#!/bin/bash
# round 4, third GPU call: eigensolver A/B (slot-polling panel kernel, four-way split, pipelined chase, T factor), the tests of the
# changed areas (eigensolver, two-rank collective decomposition, file workflows, bench launch), the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_3; mkdir -p $OUT
run() { echo "== $*" >> $OUT/eigh.txt; env "$@" GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $N >> $OUT/eigh.txt 2>&1; }
N=20000
run A=default
run GEMMA_HIP_EIGH_BC_PIPE=0
run GEMMA_HIP_EIGH_SPLIT4=0
run GEMMA_HIP_EIGH_PANEL=launch
run A=default
run GEMMA_HIP_EIGH_BC_DBG=1
N=8192; run GEMMA_HIP_EIGH_STAGES=2
N=32768; run A=default
N=50000; run A=default
grep -E "==|eigh|dense|chase" $OUT/eigh.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eigh -o e -- python scripts/eigh_probe.py 20000 > $OUT/prof_eigh.log 2>&1
find $OUT/prof_eigh -name "*kernel_stats.csv" -exec cp {} $OUT/eigh_kernel_stats.csv \;
find $OUT/prof_eigh -name "*kernel_trace.csv" -exec gzip -9 {} \;
head -16 $OUT/eigh_kernel_stats.csv | cut -c1-150
timeout 1500 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_two_rank.py tests/test_gpu_workflow_files.py tests/test_gpu_bench_launch.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -15 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_3/bench.jsonl').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['stage_ms_per_step'])
cb=l['cpu_baseline']; print({k:cb[k] for k in cb if k.startswith('gpu_vs')})
print(l['config']['setup'].get('eigen_s'), l['config']['setup'].get('eigen_stages_s'), l['roofline']['frac'], l.get('setup_parity',{}).get('eigh_resid'))
print(l['amdahl'])
PY
