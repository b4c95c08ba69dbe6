#!/bin/bash
# kernel timeline of the one-submission stereo frame (tools/c/latency_pair.c, path C)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_lat; rm -rf $OUT; mkdir -p $OUT; cd $R
gcc -O2 -I include tools/c/latency_pair.c -o /tmp/latency_pair -ldl -lpthread -lm || exit 1
GFO_DEBUG_PLAN=1 /tmp/latency_pair gf-orb-slam2_amd/libgfo.so tests/golden 5 2>&1 | grep "gfo\]" | sort | uniq -c
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- /tmp/latency_pair gf-orb-slam2_amd/libgfo.so tests/golden 20 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','trace_lat')
rows=[]
for f in glob.glob(os.path.join(root,'**','*kernel_trace.csv'),recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:36]))
for f in glob.glob(os.path.join(root,'**','*memory_copy_trace.csv'),recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')[:24]+' '+r.get('Bytes','')))
rows.sort()
# last frame-path submission: find the last 'k_stereo_cut' and walk back to the preceding H2D
idx=[i for i,r in enumerate(rows) if 'k_stereo_cut' in r[2]][-1]
start=idx
while start>0 and not ('COPY' in rows[start][2] and 'HOST_TO_DEVICE' in rows[start][2].upper().replace(' ','_')): start-=1
t0=rows[start][0]
for s,e,n in rows[start:idx+12]:
    print(f"{(s-t0)/1e3:8.1f} us  +{(e-s)/1e3:7.1f} us  {n}")
PY
