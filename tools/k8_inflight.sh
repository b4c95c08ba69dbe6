#!/bin/bash
# Same-box A/B of the combiner's batches in flight at K = 8 / 16 (through gpurun): tools/k8_inflight.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for inf in 2 3 4; do
  GFO_COMBINE_INFLIGHT=$inf bash tools/boundary_throughput.sh k8_$inf 2 both 8,16 1 0 > /dev/null 2>&1
  python3 - $inf gpurun_out/boundary_throughput_k8_$inf.json <<'PY'
import json,sys
j=json.load(open(sys.argv[2]))
print('inflight', sys.argv[1], [(p['path'][:8], p['streams'], round(p['images_per_s']), round(p['frames_per_device_batch'], 2), p['latency_ms']['p50']) for p in j['points']], flush=True)
PY
done
done
