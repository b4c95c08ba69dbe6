#!/bin/bash
# Upper bound of what fusing the blur into the pyramid kernel could gain: an instrumented build whose k_blur reads the
# SAME two images' levels for every image of the batch (cache-resident reads, identical instruction stream, wrong
# results) against the product build, same box, kernel-trace durations of k_blur.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/blur_bound; rm -rf $OUT; mkdir -p $OUT; cd $R
D=/tmp/gfo_dbg; rm -rf $D; mkdir -p $D/pkg; cp -r $R/include $D/include; cp -r $R/gf-orb-slam2_amd/csrc $D/pkg/csrc
( cd $D/pkg/csrc && rm -f *.o && make -s EXTRA=-DGFO_BLUR_DEBUG OUT=/tmp/libgfo_dbg.so ) || exit 1
for v in product cached_reads; do
  if [ $v = cached_reads ]; then export GFO_LIB=/tmp/libgfo_dbg.so; else unset GFO_LIB; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --streams 1 --workload extract752 > $OUT/$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','blur_bound')
for v in ('product','cached_reads'):
    f=glob.glob(os.path.join(root,v,'**','*kernel_trace.csv'),recursive=True)[0]
    d={}
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(v, ' '.join(f"{k}={sorted(x)[len(x)//2]:.1f}us" for k,x in sorted(d.items()) if k.startswith('k_')))
PY
