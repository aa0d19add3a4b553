#!/bin/bash
# round 4, final GPU call (after the chase hand-overs and the run-time multivariate kernel): eigensolver sanity, the full GPU suite, the driver's bench command, and
# rocprofv3 --kernel-trace --stats of the same command (setup restored from --state-file so that the trace holds the timed region)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_14; mkdir -p $OUT
GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 8192 kin > $OUT/eigh.txt 2>&1
GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 20000 kin >> $OUT/eigh.txt 2>&1
grep -E "eigh" $OUT/eigh.txt
if ! grep -q "eigh n=20000 (kin)" $OUT/eigh.txt; then echo "eigensolver failed: stopping"; tail -20 $OUT/eigh.txt; exit 3; fi
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -4 $OUT/pytest_gpu.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS --state-file /tmp/bench_state.pt > $OUT/bench_plain.jsonl 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o b -- python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS --state-file /tmp/bench_state.pt > $OUT/bench_profiled.jsonl 2> $OUT/prof_bench.log
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
rm -rf $OUT/prof_bench
head -12 $OUT/bench_kernel_stats.csv | cut -c1-140
python - <<'PY'
import json
for f in ("bench_driver_cmd","bench_plain","bench_profiled"):
    try:
        l=json.loads(open('gpurun_out/r4_14/%s.jsonl'%f).read().strip().splitlines()[-1])
        print(f, l['value'], l['ms_per_step'], l['stage_ms_per_step']['utx_gemm'], l['roofline']['frac'], l['config']['setup'].get('eigen_s'))
    except Exception as e: print(f, 'ERR', e)
l=json.loads(open('gpurun_out/r4_14/bench_driver_cmd.jsonl').read().strip().splitlines()[-1])
cb=l['cpu_baseline']; print({k:cb[k] for k in cb if k.startswith('gpu_vs')}); print(l.get('setup_parity',{}).get('eigh_resid'), l.get('c4_leg',{}).get('value'))
PY
