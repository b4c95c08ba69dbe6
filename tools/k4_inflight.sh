cd $GRAFT_REPO_ROOT
for inf in 2 1 3; do
  GFO_COMBINE_INFLIGHT=$inf bash tools/boundary_throughput.sh k4_$inf 2 stereo 2,4,6 1 1 > /dev/null 2>&1
  python - $inf gpurun_out/boundary_throughput_k4_$inf.json <<'PY'
import json,sys
j=json.load(open(sys.argv[2]))
print('inflight', sys.argv[1], [(p['streams'], p['images_per_s'], p['frames_per_device_batch'], p['latency_ms']['p50']) for p in j['points']])
PY
done
