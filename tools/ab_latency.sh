#!/bin/bash
# Same-box A/B of per-frame latency (tools/c/latency_pair.c) and of the K-stream harness between library builds.
# usage (through gpurun): tools/ab_latency.sh <libA.so> <libB.so> ...   (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-.}; cd $R; mkdir -p gpurun_out
gcc -O2 -I include tools/c/latency_pair.c -o /tmp/latency_pair -ldl -lpthread -lm || exit 1
gcc -O2 -I include tools/c/boundary_throughput.c -o /tmp/boundary_throughput -ldl -lpthread -lm || exit 1
for rep in 1 2 3; do
  for LIB in "$@"; do
    /tmp/latency_pair $LIB tests/golden 300 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[$LIB] frame %.4f  batched %.4f  adapter %.4f ms' % (j['frame_path_one_submission_ms']['median'], j['batched_path_one_context_ms']['median'], j['adapter_path_two_contexts_two_threads_ms']['median']))"
  done
done
for LIB in "$@"; do
  /tmp/boundary_throughput $LIB tests/golden 1.5 both 1,4 1 0 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read())
print('[$LIB] ' + '  '.join('%s K=%d %d img/s p50 %.4f' % (p['path'][:7], p['streams'], p['images_per_s'], p['latency_ms']['p50']) for p in j['points']))"
done
