#!/bin/bash
# How well do the contexts' kernels overlap?  Kernel trace of the bench's timed loop with N contexts, then the
# concurrency profile of the steady part: wall time, sum of kernel durations, time with 0/1/2/3+ kernels in flight,
# per-kernel average duration (to compare with the one-context durations in profiles/kernel_stats_*).
# usage (through gpurun): tools/overlap_trace.sh <tag> [streams] [extra bench args]  -> gpurun_out/overlap_<tag>.txt
cd /tmp && export TMPDIR=/tmp
TAG=$1; NS=${2:-3}; shift; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/overlap_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 60 --warmup 30 --streams $NS --no-cpu-baseline --no-other-configs --profile-steps 0 "$@" > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
python3 - "$OUT" "$NS" <<'PY' | tee $R/gpurun_out/overlap_$TAG.txt
import csv, glob, os, sys, collections
root, ns = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:40], r.get('Stream_Id', r.get('Queue_Id', '')))
        for r in csv.DictReader(open(f))]
rows.sort()
# the timed loop = the longest run of kernels without a gap above 2 ms; take its middle half (steady state)
runs, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-8:]) > 2_000_000:
        runs.append(cur); cur = []
    cur.append(r)
runs.append(cur)
run = max(runs, key=len)
t_lo = run[0][0] + (run[-1][1] - run[0][0]) // 4
t_hi = run[0][0] + 3 * (run[-1][1] - run[0][0]) // 4
ev = []
per = collections.defaultdict(list)
for s, e, k, q in run:
    if e <= t_lo or s >= t_hi: continue
    per[k].append(e - s)
    ev.append((max(s, t_lo), 1)); ev.append((min(e, t_hi), -1))
ev.sort()
depth, last, hist = 0, t_lo, collections.Counter()
for t, d in ev:
    hist[min(depth, 4)] += t - last
    last = t; depth += d
hist[min(depth, 4)] += t_hi - last
wall = t_hi - t_lo
nfast = len(per.get('k_fast<48, 44, true>', per.get('k_fast<64, 60, true>', [])))
print(f"contexts {ns}: steady window {wall/1e6:.3f} ms, {nfast} steps inside -> {wall/1e3/max(nfast,1):.1f} us per step")
print(f"sum of kernel durations / wall = {sum(sum(v) for v in per.values())/wall:.3f}")
for d in range(5):
    print(f"  {d}{'+' if d==4 else ' '} kernels in flight: {100*hist[d]/wall:5.1f} % of the time")
print("kernel, calls, avg_us under overlap")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k}, {len(v)}, {sum(v)/len(v)/1e3:.1f}")
PY
grep -v amdgpu.ids $OUT/log.txt | tail -3 | cut -c1-400
