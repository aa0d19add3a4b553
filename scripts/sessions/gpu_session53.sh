#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "lmm_assoc|grid_table" -d $R/gpurun_out/s53_pmc_$tag -o pmc -- python $R/scripts/assoc_probe.py > $R/gpurun_out/s53_pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done
cd $R
python - <<'PY'
import csv, glob, os, collections
agg=collections.defaultdict(list)
for d in sorted(glob.glob('gpurun_out/s53_pmc_*/')):
    for r in csv.DictReader(open(os.path.join(d,'pmc_counter_collection.csv'))):
        k=r['Kernel_Name'].split('(')[0]
        dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        agg[(k,r['Counter_Name'])].append((float(r['Counter_Value']),dur))
for k,v in sorted(agg.items()):
    print(k, len(v), sum(x for x,_ in v)/len(v), sum(d for _,d in v)/len(v))
PY
