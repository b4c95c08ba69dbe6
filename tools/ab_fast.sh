#!/bin/bash
# Same-box A/B of two builds of the library on the synthetic stream (bench.py) and on the EuRoC batch (tools/bench_real_image.py).
# usage (through gpurun): tools/ab_fast.sh <libA> <libB> ...   ("" = the library as built)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for rep in 1 2; do
for LIB in "$@"; do
  if [ -n "$LIB" ]; then export GFO_LIB=$R/$LIB; else unset GFO_LIB; fi
  python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic > gpurun_out/abf.json 2> gpurun_out/abf.err || { tail -5 gpurun_out/abf.err; exit 1; }
  python - "${LIB:-as built}" gpurun_out/abf.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = j["roofline"]["stage_ms_per_step"]
print(f"[{sys.argv[1]}] synthetic: value {j['value']:.0f} sustained {j.get('sustained', {}).get('value')}  " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items()))
PY
  python tools/bench_real_image.py 2> gpurun_out/abf_real.err | sed "s#^#[${LIB:-as built}] #"
done
done
