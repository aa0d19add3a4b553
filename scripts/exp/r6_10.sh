#!/bin/bash
# Round 6, lease 9: (1) the whole GPU suite on the round's library; (2) bench.py --gpus N at the headline size through the real multi-rank flow
# behind the staged start, N ranks on the ONE device of the box (the driver's default process group, the library's shm test transport:
# RCCL refuses two ranks on one device) -- the protocol at full size; the timings share one GPU and mean nothing; (3) the same with a
# failure injected into the first large collective; (4) rocprofv3 kernel stats of the eigensolver at n = 50 000
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_all.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/pytest_gpu_all.txt | tail -12
for spec in "2:" "4:" "2:allreduce_large"; do
  N=${spec%%:*}; F=${spec#*:}
  GEMMA_HIP_COMM_TIMING=1 GEMMA_HIP_COMM_FAIL=$F BENCH_FORCE_DEVICE=0 GEMMA_HIP_COMM=shm timeout 900 python bench.py --gpus $N --steps 4 --warmup 1 --cpu-sample 0 > $OUT/bench_${N}ranks_shm_${F:-ok}.jsonl 2> $OUT/bench_${N}ranks_shm_${F:-ok}.err; echo "N=$N fail=$F rc=$?"
  python - $OUT/bench_${N}ranks_shm_${F:-ok}.jsonl <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = l["config"]["setup"]; c = l["config"]["comm"]
    print(" ", l["n_gpus"], l["value"], l["ms_per_step"], "ranks_seen", l["config"]["ranks_seen"], l["config"]["per_rank"]["value"])
    print("  comm:", {k: c.get(k) for k in ("setup_mode", "control_plane", "error", "collectives")})
    print("  staged_start:", [(t["stage"][:50], t["ok"], t.get("seconds")) for t in c["staged_start"]])
    print("  setup:", {k: s.get(k) for k in ("kinship_s", "allreduce_s", "eigen_workspace_reserve_s", "eigen_s", "broadcast_s", "broadcast")})
except Exception as e:
    print("no line", repr(e))
PY
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eigh -o e -- python scripts/eigh_probe.py 50000 kin > $OUT/eigh_n50000_prof.txt 2>&1
find $OUT/prof_eigh -name "*kernel_stats.csv" -exec cp {} $OUT/eigh_n50000_kernel_stats.csv \;
rm -rf $OUT/prof_eigh
grep -E "eigh" $OUT/eigh_n50000_prof.txt | tail -3; head -12 $OUT/eigh_n50000_kernel_stats.csv | cut -c1-160
