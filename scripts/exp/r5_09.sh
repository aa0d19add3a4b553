#!/bin/bash
# per-kernel split of the eigensolver at n = 50000 (kinship-like spectrum), then the plain timing with the residual check
GEMMA_HIP_EIGH_TIMING=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o e -- python scripts/eigh_probe.py 50000 kin > $OUT/eigh50k_profiled.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eigh50k_kernel_stats.csv \;
python - <<'PY'
import csv, glob, os, collections
out = os.environ["OUT"]
# GEMM launches by shape class: duration histogram of dgemm kernels from the trace (grid size tells the shape)
for f in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "dgemm" in k:
            key = (k[-40:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            a = acc[key]; a[0] += 1; a[1] += d
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])[:0]
    # too many shapes: aggregate by kernel and by decile of duration instead
    agg = collections.defaultdict(lambda: [0, 0.0])
    for (k, gx, gy), (c, t) in acc.items():
        agg[k][0] += c; agg[k][1] += t
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-42s launches %6d total %9.1f ms" % (k, c, t))
PY
rm -rf $OUT/prof
grep -E "eigh" $OUT/eigh50k_profiled.txt
head -22 $OUT/eigh50k_kernel_stats.csv | cut -c1-160
EIGH_PROBE_CHECK=1 GEMMA_HIP_EIGH_TIMING=1 timeout 300 python scripts/eigh_probe.py 50000 kin > $OUT/eigh50k_plain.txt 2>&1; grep -E "eigh" $OUT/eigh50k_plain.txt
