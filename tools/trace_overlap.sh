#!/bin/bash
# kernel-trace of the multi-context bench: how much of the wall time has 0 / 1 / 2+ kernels in flight
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_ov; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','trace_ov')
f=glob.glob(os.path.join(root,'**','*kernel_trace.csv'),recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//3:]           # steady state
ev=[]
for r in rows:
    ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort()
t_prev=ev[0][0]; depth=0; hist=collections.Counter()
for t,d in ev:
    hist[min(depth,3)]+=t-t_prev; t_prev=t; depth+=d
tot=sum(hist.values())
print('span %.1f us; in flight: '%(tot/1e3)+', '.join(f'{k}: {100*v/tot:.1f}%' for k,v in sorted(hist.items())))
per=collections.defaultdict(list)
for r in rows: per[r['Kernel_Name'].split('(')[0][:28]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(per.items(), key=lambda kv:-sum(kv[1])): print(f'{k:30s} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.1f} us')
PY
