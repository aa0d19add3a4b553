#!/bin/bash
# the two-block pipeline inside the library: its equality test, then the driver's timed region with and without it
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0 --kin-snps 20000 --state-file /tmp/bench_state.pt"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipe_blocks or reload_env or variants_agree" > $OUT/pipe_tests.txt 2>&1; tail -5 $OUT/pipe_tests.txt
for CUS in 64 32; do
GEMMA_HIP_PIPE_CUS=$CUS timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $OUT/bench_pipe_$CUS.jsonl 2> $OUT/bench_pipe_$CUS.err; echo "pipe $CUS rc=$?"
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pipeline 0 $LEGS > $OUT/bench_nopipe.jsonl 2> $OUT/bench_nopipe.err; echo "nopipe rc=$?"
python - <<'PY'
import json, os
for f in ("bench_pipe_64", "bench_pipe_32", "bench_nopipe"):
    try:
        l = json.loads(open(os.environ["OUT"] + "/" + f + ".jsonl").read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], {k: v for k, v in l["stage_ms_per_step"].items() if k != "overlap"}, l["roofline"]["frac"], l.get("one_stream_leg"))
    except Exception as e:
        print(f, "no line", repr(e)); os.system("tail -5 %s/%s.err" % (os.environ["OUT"], f))
PY
