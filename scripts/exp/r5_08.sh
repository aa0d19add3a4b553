#!/bin/bash
# two blocks in flight: which partition (if any) pays?  mask bit c = CU c / 8 of XCD c % 8 (r5_07), a mask that empties an XCD is not applied
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0 --kin-snps 20000 --state-file /tmp/bench_state.pt"
timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --pipeline 0 $LEGS > /dev/null 2>&1
for rep in 1 2; do
for CUS in nopipe 0 32 64 128; do
  if [ $CUS = nopipe ]; then A="--pipeline 0"; E=64; else A=""; E=$CUS; fi
  GEMMA_HIP_PIPE_CUS=$E timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $A $LEGS > $OUT/b_${CUS}_$rep.jsonl 2> $OUT/b_${CUS}_$rep.err
  python - $OUT/b_${CUS}_$rep.jsonl $CUS <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-7s" % sys.argv[2], l["value"], l["ms_per_step"], {k: v for k, v in l["stage_ms_per_step"].items() if k != "overlap"})
PY
done
done
