#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for cfg in "8 4" "8 16" "8 2" "4 4" "4 16"; do
  set -- $cfg
  GEMMA_HIP_GEMM_WAVES=$1 GEMMA_HIP_GEMM_GM=$2 GEMMA_HIP_GEMM_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv --kernel-include-regex "dgemm_mfma" -d $GRAFT_REPO_ROOT/gpurun_out/pmc4_w$1_g$2 -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > $GRAFT_REPO_ROOT/gpurun_out/pmc4.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for d in sorted(glob.glob('gpurun_out/pmc4_w*')):
    rows=list(csv.DictReader(open(d+'/pmc_counter_collection.csv')))
    acc={}
    for r in rows:
        if int(r['Grid_Size'])>=156*156*256 and 'true>' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
            dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    h=sum(acc['TCC_HIT_sum'])/len(acc['TCC_HIT_sum']); m=sum(acc['TCC_MISS_sum'])/len(acc['TCC_MISS_sum'])
    print(d, 'hit rate %.3f'%(h/(h+m)), 'last dur %.1f ms'%dur)
PY
