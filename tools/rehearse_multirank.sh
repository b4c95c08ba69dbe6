#!/bin/bash
# Rehearses bench.py's N > 1 path on a ONE-GPU box: N ranks (default 2) share device 0 and talk over gloo
# (GFO_BENCH_REHEARSAL=1).  Checks the control flow the driver's multi-GPU run takes: rendezvous, per-rank streams,
# the count all-gather every step, barriers, max-over-ranks timing, rank 0's single JSON line.  Not a measurement.
N=${1:-2}
cd ${GRAFT_REPO_ROOT:-.}
GFO_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus $N --steps 10 --warmup 2 --batch 32 2> gpurun_out/rehearse.err | tee gpurun_out/rehearse.json | python -c "
import json,sys
lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
assert len(lines)==1, lines
d=json.loads(lines[0]); print('n_gpus', d['n_gpus'], 'value', d['value'], 'scaling', d['scaling'], 'sharding', d['config']['sharding'], 'kernel', d['roofline']['kernel'])"
echo rc=$?
tail -3 gpurun_out/rehearse.err
