#!/bin/bash
# Round 6: counters of the fp64 MFMA GEMM path (U^T x as ONE dgemm_mfma_glds_kernel launch per block; GEMMA_HIP_UTX_I8=0) -- the
# utx_gemm_* entries of profiles/pmc_traffic.json were a round-1 figure.  One rocprofv3 --pmc pass per counter group,
# setup restored from the state file scripts/pmc_bench.sh leaves behind (run the `pmc` step first).
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0 --kin-snps 20000 --state-file /tmp/bench_state.pt"
export GEMMA_HIP_UTX_I8=0
mkdir -p $OUT/pmc_fp64
[ -f /tmp/bench_state.pt ] || python bench.py $ARGS > $OUT/pmc_fp64/setup.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-include-regex "dgemm_mfma" --kernel-trace --output-format csv -d "$OUT/pmc_fp64/pass$i" -o p -- \
      python bench.py $ARGS > "$OUT/pmc_fp64/pass$i.log" 2>&1
  echo "fp64 pass $i ($C): rc=$?"
done
python3 - "$OUT/pmc_fp64" <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
acc = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        # the U^T x launch of a block is the big one: grid of (B / 128) x (n / 128) workgroups; small launches (table products' helpers) are left out
        if int(r.get("Grid_Size", 0) or 0) >= 256 * 20000:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pass*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d_ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if "dgemm_mfma" in r["Kernel_Name"] and d_ms > 100.0:  # the block's U^T x launch (~220 ms); everything else is far shorter
            dur.append(d_ms)
m = {k: sum(v) / len(v) for k, v in acc.items()}
ms = sum(dur) / len(dur) if dur else None
res = {"utx_gemm_kernel": "dgemm_mfma_glds_kernel", "launches_counted": {k: len(v) for k, v in acc.items()},
       "utx_gemm_hbm_bytes_per_launch": round(2048.0 * m.get("FETCH_SIZE", 0) + 1024.0 * m.get("WRITE_SIZE", 0)),
       "utx_gemm_tcc_hit_rate": round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in m else None,
       "utx_gemm_launch_ms_under_counters": round(ms, 3) if ms else None,
       "utx_gemm_clock_GHz": round(m["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e6), 3) if ms and "GRBM_GUI_ACTIVE" in m else None,
       "utx_gemm_mfma_util": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0), 4) if "GRBM_GUI_ACTIVE" in m else None}
print(json.dumps(res, indent=1))
open(out + "/fp64_gemm_counters.json", "w").write(json.dumps(res, indent=1))
PY
find $OUT/pmc_fp64 -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
