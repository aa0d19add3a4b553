#!/bin/bash
# Lease 20: complete-block form of the records product -- the new parity test, then the plain step on complete blocks at n = B = 20 000
# (bench.py complete_leg) with the kernel trace of that run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/r6_20}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "complete_blocks or strict_form or side_stream or sparse_mask" > $OUT/test_complete.txt 2>&1; tail -5 $OUT/test_complete.txt
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python bench.py --gpus 1 --steps 10 --warmup 3 --complete-steps 6 $LEGS > $OUT/bench_complete.jsonl 2> $OUT/bench_complete.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_complete_kernel_stats.csv \;
rm -rf $OUT/prof
head -12 $OUT/bench_complete_kernel_stats.csv | cut -c1-200
python - <<PY
import json
l = json.loads(open("$OUT/bench_complete.jsonl").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["frac"])
print(json.dumps(l.get("complete_leg"), indent=1))
PY
