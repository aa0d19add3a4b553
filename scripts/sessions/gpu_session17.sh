#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for w in 4 8; do
  GEMMA_HIP_GEMM_WAVES=$w GEMMA_HIP_GEMM_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma" -d $GRAFT_REPO_ROOT/gpurun_out/pmc3_w$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > $GRAFT_REPO_ROOT/gpurun_out/pmc3_w$w.log 2>&1
  GEMMA_HIP_GEMM_WAVES=$w GEMMA_HIP_GEMM_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma" -d $GRAFT_REPO_ROOT/gpurun_out/pmc3h_w$w -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > $GRAFT_REPO_ROOT/gpurun_out/pmc3h_w$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for d in ("pmc3_w4","pmc3_w8","pmc3h_w4","pmc3h_w8"):
    try:
        rows=list(csv.DictReader(open('gpurun_out/%s/pmc_counter_collection.csv'%d)))
    except Exception as e:
        print(d, 'missing', e); continue
    for r in rows:
        if int(r['Grid_Size'])>=156*156*256 and 'true>' in r['Kernel_Name']:
            print(d, r['Counter_Name'], r['Counter_Value'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
