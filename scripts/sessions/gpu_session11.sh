#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for w in 8 4; do
GEMMA_HIP_GEMM_WAVES=$w timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_w$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_w$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections
for w in (8,4):
    rows=list(csv.DictReader(open('gpurun_out/pmc_gemm_w%d/pmc_counter_collection.csv'%w)))
    agg=collections.defaultdict(list)
    for r in rows:
        if int(r['Grid_Size'])>=156*156*256: agg[(r['Kernel_Name'][17:58], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(w, k, '%.4g'%(sum(v)/len(v)), len(v))
PY
