#!/bin/bash
# Where does the K-stream per-frame path saturate?  Kernel + copy + HIP API trace of tools/c/boundary_throughput.c at one K:
# kernels in flight over time, device time per frame, and what every HIP call costs the host threads.
# usage (through gpurun): tools/trace_boundary.sh <tag> <K> [mode] [api:0|1] [combine:0|1]  -> gpurun_out/trace_boundary_<tag>.txt
cd /tmp && export TMPDIR=/tmp
TAG=$1; K=${2:-8}; MODE=${3:-stereo}; API=${4:-1}; CMB=${5:-1}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_boundary_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
gcc -O2 -I include tools/c/boundary_throughput.c -o /tmp/boundary_throughput -ldl -lpthread -lm || exit 1
EXTRA=""; [ "$API" = "1" ] && EXTRA="--hip-runtime-trace"
GFO_DUMP_MAPS=1 timeout -k 10 240 rocprofv3 --kernel-trace --memory-copy-trace $EXTRA --output-format csv -d $OUT -- /tmp/boundary_throughput gf-orb-slam2_amd/libgfo.so tests/golden 0.5 $MODE $K $CMB > $OUT/log.txt 2>&1
grep -v amdgpu.ids $OUT/log.txt | tail -4 | cut -c1-330
python3 - "$OUT" <<'PY' | tee $R/gpurun_out/trace_boundary_$TAG.txt
import csv, glob, os, sys, collections
root = sys.argv[1]
def load(pat):
    fs = glob.glob(os.path.join(root, '**', pat), recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
kern = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:36]) for r in load('*kernel_trace.csv')]
cop = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')[:20]) for r in load('*memory_copy_trace.csv')]
kern.sort()
t_hi = kern[-1][1]; t_lo = t_hi - 300_000_000   # the last 0.3 s: steady state of the timed region
win = [k for k in kern if k[0] >= t_lo]
ev = []
for s, e, n in win: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t_lo, collections.Counter()
for t, d in ev:
    hist[min(depth, 8)] += t - last; last = t; depth += d
wall = t_hi - t_lo
per = collections.defaultdict(list)
for s, e, n in win: per[n].append(e - s)
frames = len(per.get("k_stereo_cut", [])) or len(per.get("k_pack_results_cut", [])) or 1   # (round 5: the per-frame path has no k_stereo_cut launch; the pack kernel makes the cut)
print(f"window {wall/1e6:.1f} ms, {frames} stereo frames -> {wall/1e3/frames:.1f} us per frame; sum of kernel durations per frame {sum(sum(v) for v in per.values())/1e3/frames:.1f} us")
for d in range(9): print(f"  {d}{'+' if d==8 else ' '} kernels in flight: {100*hist[d]/wall:5.1f} %")
print("kernel, launches per frame, avg us")
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])): print(f"  {n}, {len(v)/frames:.2f}, {sum(v)/len(v)/1e3:.1f}")
cw = [c for c in cop if c[0] >= t_lo]
cper = collections.defaultdict(list)
for s, e, n in cw: cper[n].append(e - s)
for n, v in cper.items(): print(f"  {n}: {len(v)/frames:.2f} per frame, avg {sum(v)/len(v)/1e3:.1f} us")
api = load('*hip_api_trace.csv')
if api:
    a = collections.defaultdict(list)
    for r in api:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s >= t_lo: a[r['Function']].append(e - s)
    tot = sum(sum(v) for v in a.values())
    print(f"HIP API time per frame (summed over host threads): {tot/1e3/frames:.1f} us")
    for n, v in sorted(a.items(), key=lambda kv: -sum(kv[1]))[:14]: print(f"  {n}: {len(v)/frames:.2f} calls per frame, avg {sum(v)/len(v)/1e3:.2f} us, per frame {sum(v)/1e3/frames:.1f} us")
PY
