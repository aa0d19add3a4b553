#!/bin/bash
# is the CU mask of hipExtStreamCreateWithCUMask honoured inside the python process (torch's bundled HIP runtime)?
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0 --kin-snps 20000 --state-file /tmp/bench_state.pt"
TL=$(python -c "import torch,os;print(os.path.dirname(torch.__file__)+'/lib')")
{
echo "== harness, system runtime, split 64"; CU_SPLIT=64 CU_MODE=1 RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb13 20000 20000 7 0
echo "== harness, torch's runtime ($TL), split 64"; LD_LIBRARY_PATH=$TL CU_SPLIT=64 CU_MODE=1 RASTER=1 REPS=4 timeout 60 scripts/abl_bin/kb13 20000 20000 7 0
echo "== harness, torch's runtime, split 128"; LD_LIBRARY_PATH=$TL CU_SPLIT=128 CU_MODE=1 RASTER=1 REPS=2 timeout 60 scripts/abl_bin/kb13 20000 20000 7 0
echo "== harness, system runtime, split 128"; CU_SPLIT=128 CU_MODE=1 RASTER=1 REPS=2 timeout 60 scripts/abl_bin/kb13 20000 20000 7 0
} > $OUT/mask_runtime.txt 2>&1
grep -E "==|variant|CU_SPLIT|rror" $OUT/mask_runtime.txt
GEMMA_HIP_PIPE_CUS=128 timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 $LEGS > $OUT/bench_pipe_128.jsonl 2> $OUT/bench_pipe_128.err; echo "pipe 128 rc=$?"
python - <<'PY'
import json, os
l = json.loads(open(os.environ["OUT"] + "/bench_pipe_128.jsonl").read().strip().splitlines()[-1])
print("PIPE_CUS=128:", l["value"], l["ms_per_step"], {k: v for k, v in l["stage_ms_per_step"].items() if k != "overlap"})
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipe_blocks" > $OUT/pipe_tests.txt 2>&1; tail -3 $OUT/pipe_tests.txt
