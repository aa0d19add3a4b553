#!/bin/bash
# lease 7: the round's new GPU tests, then BASELINE config 4 with one flag on one GPU (--config 4: n = 50000, p = 500000, 25 steps)
timeout 1500 python -m pytest tests -m gpu -q -k "panel_groups or workspace_pool or strict_form or side_stream or sharded_backtransformation or xlarge" --durations=6 > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.txt
tail -15 $OUT/pytest_new.txt
timeout 1500 python bench.py --gpus 1 --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.jsonl 2> $OUT/bench_config4.err; echo "config 4 rc=$?"
python - <<'PY'
import json, os
try:
    l = json.loads(open(os.environ["OUT"] + "/bench_config4.jsonl").read().strip().splitlines()[-1])
    s = l["config"]["setup"]
    print("config 4:", l["metric"]); print(" value", l["value"], l["unit"], "steps", l["steps"], "ms/step", l["ms_per_step"], "scaling", l["scaling"], "frac", l["roofline"]["frac"])
    print(" workload:", l["config"]["workload"][:200])
    print(" setup:", {k: s.get(k) for k in ("kinship_s", "eigen_workspace_reserve_s", "eigen_s", "eigen_stages_s", "setup_total_s")})
    cb = l.get("cpu_baseline", {}); print(" vs reference:", cb.get("gpu_vs_reference_max_rel_err"), cb.get("gpu_vs_reference_lambda"), cb.get("gpu_vs_oracle_max_rel_err"))
    print(" amdahl:", l["amdahl"]["projected_total_s"], l["amdahl"]["serial_fraction_at_8"])
except Exception as e:
    print("no line:", repr(e)); print(open(os.environ["OUT"] + "/bench_config4.err").read()[-1500:])
PY
