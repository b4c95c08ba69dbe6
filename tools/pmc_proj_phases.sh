#!/bin/bash
# Instruction counts and duration of k_proj_round0 truncated after a phase (instrumented build, GFO_PROJ_STOP):
# 1 = query load only, 2 = + grid scan (no descriptor fetch / distances), 0 = whole kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_proj; rm -rf $OUT; mkdir -p $OUT; cd $R
D=/tmp/gfo_dbg; rm -rf $D; mkdir -p $D/pkg; cp -r $R/include $D/include; cp -r $R/gf-orb-slam2_amd/csrc $D/pkg/csrc
( cd $D/pkg/csrc && rm -f *.o && make -s EXTRA=-DGFO_PROJ_DEBUG OUT=/tmp/libgfo_dbg.so ) || exit 1
export GFO_LIB=/tmp/libgfo_dbg.so
for s in 1 2 0; do
  export GFO_PROJ_STOP=$s
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/s$s -- python3 tools/proj_diag.py 128 > $OUT/s$s.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t$s -- python3 tools/proj_diag.py 128 > $OUT/t$s.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_proj')
for s in (1,2,0):
    f=glob.glob(os.path.join(root,f's{s}','**','*counter_collection.csv'),recursive=True)
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if 'k_proj_round0' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
    w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
    t=glob.glob(os.path.join(root,f't{s}','**','*kernel_trace.csv'),recursive=True)
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(t[0])) if 'k_proj_round0' in r['Kernel_Name']]
    print('stop',s,' '.join(f"{c}/wave={sum(x)/len(x)/w:.0f}" for c,x in sorted(acc.items())), 'us', round(sorted(d)[len(d)//2],1))
PY
