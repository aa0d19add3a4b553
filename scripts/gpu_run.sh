#!/bin/bash
# The one runner of GPU leases (replaces the per-lease scratch scripts of rounds 1-4):
#   gpurun --timeout T -- 'scripts/gpu_run.sh <tag> <step> [<step> ...]'
# Everything goes to gpurun_out/<tag>/.  Steps:
#   suite      python -m pytest tests -m gpu -x -q           -> pytest_gpu.txt
#   smoke      __graft_entry__.smoke()                       -> smoke.txt
#   bench      the driver's command (--gpus 1 --steps 20 --warmup 5, every side leg)   -> bench_driver_cmd.jsonl
#   plain      the same timed region without the side legs   -> bench_plain.jsonl
#   prof       rocprofv3 --kernel-trace --stats of `plain`   -> bench_kernel_stats.csv (+ the line the profiled run printed)
#   pmc        scripts/pmc_bench.sh (one counter pass per group on the timed kernels) + scripts/pmc_summarize.py -> pmc/summary.csv, pmc_traffic.json
#   test:<k>   pytest -m gpu -k <k>                          -> test_<k>.txt
#   py:<file>[:args]   python <file> args (',' separates args)  -> <file>.txt
#   sh:<file>  bash <file> (a one-off experiment kept under scripts/exp/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
LEGS="--cpu-sample 0 --fp64-steps 0 --dosage-steps 0 --miss-leg 0 --lowh2-leg 0 --digits7-steps 0 --setup-parity 0 --c4-leg 0 --e2e-snps 0 --complete-steps 0"
summ() { python - "$1" <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = l["roofline"]
    print("value %.1f SNPs/s  ms/step %.3f  stages %s" % (l["value"], l["ms_per_step"], {k: v for k, v in l["stage_ms_per_step"].items() if k != "overlap"}))
    print("kernel %s  frac %.4f  avg_launch_ms %.3f  traffic %s" % (r.get("kernel_symbol"), r["frac"], r["avg_launch_ms"], r.get("traffic")))
    s = l["config"]["setup"]
    print("setup: kinship_s %s eigen_s %s stages %s" % (s.get("kinship_s"), s.get("eigen_s"), s.get("eigen_stages_s")))
    for k in ("dosage_path", "strict_leg", "fp64_gemm_path"):
        if k in l: print(k, l[k].get("value"), l[k].get("ms_per_step"), (l[k].get("roofline") or {}).get("frac") or l[k].get("roofline_frac"), l[k].get("error"))
    if "c4_leg" in l: print("c4_leg", l["c4_leg"].get("value"), (l["c4_leg"].get("setup") or {}).get("eigen_s"), (l["c4_leg"].get("setup") or {}).get("eigen_stages_s"))
    if "e2e" in l: print("e2e", {k: l["e2e"].get(k) for k in ("snps", "wall_s")})
    if "cpu_baseline" in l: print("vs reference", l["cpu_baseline"].get("gpu_vs_reference_max_rel_err"), l["cpu_baseline"].get("gpu_vs_reference_lambda"))
    if "setup_parity" in l: print("setup_parity", {k: v for k, v in l["setup_parity"].items() if not k.endswith("what")})
except Exception as e:
    print("no bench line:", repr(e))
PY
}
for STEP in "$@"; do
  echo "##### $STEP"
  case $STEP in
    suite) timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -25 $OUT/pytest_gpu.txt ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt ;;
    bench) timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.jsonl 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"; summ $OUT/bench_driver_cmd.jsonl ;;
    plain) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $OUT/bench_plain.jsonl 2> $OUT/bench_plain.err; echo "bench rc=$?"; summ $OUT/bench_plain.jsonl ;;
    prof)
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $OUT/bench_profiled.jsonl 2> $OUT/bench_profiled.err
      echo "rocprofv3 rc=$?"
      find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
      rm -rf $OUT/prof
      head -14 $OUT/bench_kernel_stats.csv | cut -c1-170; summ $OUT/bench_profiled.jsonl ;;
    pmc)
      bash scripts/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1; grep -E "^pass|rc=" $OUT/pmc.log
      python scripts/pmc_summarize.py $OUT/pmc > $OUT/pmc_traffic.json 2> $OUT/pmc_summarize.err; head -c 1500 $OUT/pmc_traffic.json
      find $OUT/pmc -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ;;
    test:*) K=${STEP#test:}; timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" --durations=5 > "$OUT/test_$K.txt" 2>&1; tail -8 "$OUT/test_$K.txt" ;;
    py:*) S=${STEP#py:}; F=${S%%:*}; A=""; [ "$S" != "$F" ] && A=$(echo "${S#*:}" | tr ',' ' '); timeout 900 python $F $A > "$OUT/$(basename $F).txt" 2>&1; tail -30 "$OUT/$(basename $F).txt" ;;
    sh:*) F=${STEP#sh:}; OUT=$OUT timeout 1500 bash $F 2>&1 | tail -60 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
