#!/bin/bash
# kernel + copy trace of the bench (value_with_h2d / value_delivered legs): how long do the input copy and the delivery take?
# usage (through gpurun): tools/trace_deliver.sh <tag> [GFO_DELIVER_DMA]
cd /tmp && export TMPDIR=/tmp
TAG=$1; export GFO_DELIVER_DMA=${2:-0}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_deliver_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-boundary --profile-steps 1 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY' | tee $R/gpurun_out/trace_deliver_$TAG.txt
import csv, glob, os, sys, collections
root = sys.argv[1]
def load(pat):
    fs = glob.glob(os.path.join(root, '**', pat), recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
k = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:30]) for r in load('*kernel_trace.csv')]
c = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'], int(r.get('Bytes', 0) or 0)) for r in load('*memory_copy_trace.csv')]
pk = [x for x in k if 'k_pack' in x[2]]
print("k_pack_results launches", len(pk), "avg us", sum(e - s for s, e, _ in pk) / max(1, len(pk)) / 1e3)
big = collections.defaultdict(list)
for s, e, d, b in c:
    if e - s > 100_000: big[d].append((e - s, b))
for d, v in big.items():
    print(d, len(v), "copies > 100 us, avg us", sum(x for x, _ in v) / len(v) / 1e3, "avg MB", sum(b for _, b in v) / len(v) / 1e6)
PY
grep -v amdgpu.ids $OUT/log.txt | tail -2 | cut -c1-300
