#!/bin/bash
# Kernel timeline of the per-frame path at K = 1 (through gpurun): tools/trace_frame.sh <tag> [stereo|adapter]
# -> gpurun_out/trace_frame_<tag>.txt : the last three frames kernel by kernel + per-kernel averages
cd /tmp && export TMPDIR=/tmp
TAG=$1; MODE=${2:-stereo}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_frame_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
gcc -O2 -I include tools/c/boundary_throughput.c -o /tmp/boundary_throughput -ldl -lpthread -lm || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- /tmp/boundary_throughput gf-orb-slam2_amd/libgfo.so tests/golden 0.5 $MODE 1 1 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY' | tee $R/gpurun_out/trace_frame_$TAG.txt
import csv, glob, sys, collections
root = sys.argv[1]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]) for r in csv.DictReader(open(f))]
c = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)
if c: rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[-14:]) for r in csv.DictReader(open(c[0]))]
rows.sort()
n = len(rows)
per = collections.defaultdict(list)
for s, e, k in rows[n // 2:]: per[k].append(e - s)
print("per-kernel averages over the second half of the run (us):")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])): print(f"  {k:36s} {sum(v)/len(v)/1e3:7.1f}  x{len(v)}")
t0 = rows[n - 36][0]
print("the last frames (start us, duration us):")
for s, e, k in rows[n - 36:n - 6]: print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:6.1f} {k}")
PY
