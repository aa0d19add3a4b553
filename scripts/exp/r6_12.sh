#!/bin/bash
# lease 11: counters of q2_apply_kernel alone (n = 20000): what does it wait for?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $C --kernel-include-regex "q2_apply" --kernel-trace --output-format csv -d "$OUT/q2pmc$i" -o p -- python scripts/eigh_probe.py 20000 kin > "$OUT/q2pmc$i.log" 2>&1
  echo "q2 pmc pass $i ($C): rc=$?"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/q2pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:30], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
for f in sorted(glob.glob(out + "/q2pmc1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "q2_apply" in r["Kernel_Name"]:
            print("q2 duration ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
find $OUT -mindepth 1 -maxdepth 1 -type d -name "q2pmc*" -exec rm -rf {} +
