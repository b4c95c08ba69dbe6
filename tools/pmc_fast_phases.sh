#!/bin/bash
# VALU/SALU/LDS instruction counts of k_fast truncated after each phase (GFO_FAST_STOP=1..4, 0 = whole kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fast; rm -rf $OUT; mkdir -p $OUT; cd $R
# the truncation hooks are compiled out of libgfo.so: build an instrumented copy beside it
D=/tmp/gfo_dbg; rm -rf $D; mkdir -p $D/pkg; cp -r $R/include $D/include; cp -r $R/gf-orb-slam2_amd/csrc $D/pkg/csrc
( cd $D/pkg/csrc && rm -f *.o && make -s EXTRA=-DGFO_FAST_DEBUG OUT=/tmp/libgfo_dbg.so ) || exit 1
export GFO_LIB=/tmp/libgfo_dbg.so
for s in 1 2 3 4 0; do
  export GFO_FAST_STOP=$s
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/s$s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --profile-steps 1 --streams 1 > $OUT/s$s.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_fast')
for s in (1,2,3,4,0):
    f=glob.glob(os.path.join(root,f's{s}','**','*counter_collection.csv'),recursive=True)
    if not f: print('no csv',s); continue
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if 'k_fast' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
    print('stop',s,' '.join(f"{c}/wave={sum(x)/len(x)/w:.1f}" for c,x in sorted(acc.items())))
PY
