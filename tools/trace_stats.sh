#!/bin/bash
# Kernel-trace statistics of one bench.py command (through gpurun): per-kernel calls / average / total.
# usage: tools/trace_stats.sh <tag> "<bench args>"   -> gpurun_out/trace_<tag>/stats.txt
cd /tmp && export TMPDIR=/tmp
TAG=${1:-x}; ARGS=${2:---steps 10 --warmup 2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py $ARGS --no-cpu-baseline --no-other-configs > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root=sys.argv[1]
f=glob.glob(os.path.join(root,'**','*kernel_trace.csv'),recursive=True)[0]
per=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    per[r['Kernel_Name'].split('(')[0].replace('void ','')[:60]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=sum(sum(v) for v in per.values())
lines=["kernel,calls,total_ms,avg_us,min_us,max_us,share"]
for k,v in sorted(per.items(), key=lambda kv:-sum(kv[1])):
    lines.append(f"{k},{len(v)},{sum(v)/1e6:.3f},{sum(v)/len(v)/1e3:.2f},{min(v)/1e3:.2f},{max(v)/1e3:.2f},{100*sum(v)/tot:.1f}%")
open(os.path.join(root,'stats.txt'),'w').write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
tail -1 $OUT/log.txt | head -c 600
