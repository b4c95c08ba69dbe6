#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_q; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --profile-steps 1 ${BENCH_ARGS} > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','trace_q')
f=glob.glob(os.path.join(root,'**','*kernel_trace.csv'),recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-40:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')} wg {r.get('Workgroup_Size_X','?')} {r['Kernel_Name'].split('(')[0][:40]}")
PY
