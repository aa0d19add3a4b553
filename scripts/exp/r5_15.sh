#!/bin/bash
# bench.py --gpus N at the headline size through the real multi-rank flow, N ranks on the ONE device of the box over the library's shm
# test transport (RCCL refuses two ranks on one device): the protocol at full size; the timings share one GPU and mean nothing
for N in 2 4; do
  BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo GEMMA_HIP_COMM=shm timeout 900 python bench.py --gpus $N --steps 4 --warmup 1 --cpu-sample 0 > $OUT/bench_${N}ranks_shm.jsonl 2> $OUT/bench_${N}ranks_shm.err; echo "N=$N rc=$?"
  python - $OUT/bench_${N}ranks_shm.jsonl <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = l["config"]["setup"]
    print(l["n_gpus"], l["value"], l["ms_per_step"], l["config"]["ranks_seen"], l["config"]["per_rank"], l["config"]["comm"]["transport"])
    print({k: s.get(k) for k in ("flow", "kinship_s", "kinship_snps_per_rank", "kinship_snps_all_ranks", "allreduce_s", "eigen_s", "eigen", "broadcast_s", "slowest_rank_s")})
    print("nan_p_wald", l["config"]["nan_p_wald"], "amdahl", l["amdahl"]["projected_total_s"])
except Exception as e:
    print("no line", repr(e))
PY
  tail -3 $OUT/bench_${N}ranks_shm.err
done
