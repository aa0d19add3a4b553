#!/bin/bash
# BASELINE configs 2 and 5 on the round's last library (parity-test cases, not bench lines: for the record)
timeout 600 python bench.py --individuals 5000 --a-mode 4 --kin-snps 100000 --steps 10 --warmup 2 --cpu-sample 512 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --c4-leg 0 --e2e-snps 0 > $OUT/bench_c2_n5000_lmm4.jsonl 2> $OUT/bench_c2.err; echo "c2 rc=$?"
python - <<'PY'
import json, os
l = json.loads(open(os.environ["OUT"] + "/bench_c2_n5000_lmm4.jsonl").read().strip().splitlines()[-1])
print("config 2:", l["value"], "SNPs/s", l["ms_per_step"], "ms/step", {k: v for k, v in l["stage_ms_per_step"].items() if k != "overlap"}, l["roofline"]["kernel_symbol"], l["roofline"]["frac"])
print("  vs reference:", l.get("cpu_baseline", {}).get("gpu_vs_reference_max_rel_err"), l.get("cpu_baseline", {}).get("gpu_vs_oracle_max_rel_err"))
PY
timeout 600 python scripts/mvlmm_probe.py 10000 16384 3 4 > $OUT/mvlmm_c5.txt 2>&1; grep -E "mvlmm batch|null block|oracle|max" $OUT/mvlmm_c5.txt
