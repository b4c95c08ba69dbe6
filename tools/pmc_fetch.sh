#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_f; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --profile-steps 1 --streams 1 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_f')
f=glob.glob(os.path.join(root,'**','*counter_collection.csv'),recursive=True)[0]
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    acc[row["Kernel_Name"].split("(")[0].replace("void ","")].append(float(row['Counter_Value']))
for k,v in acc.items():
    if k.startswith('k_'): print(k, round(sum(v)/len(v)/1024,1), 'MB/launch')
PY
