#!/bin/bash
# Same-box A/B of environment settings: tools/ab_env.sh "VAR=val VAR2=val" "" ...   ("" = defaults); two short bench runs each
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for rep in 1 2; do
for E in "$@"; do
  env $E python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic ${GFO_AB_ARGS} > gpurun_out/abe.json 2> gpurun_out/abe.err || { tail -5 gpurun_out/abe.err; exit 1; }
  python - "${E:-defaults}" gpurun_out/abe.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = j["roofline"]["stage_ms_per_step"]
print(f"[{sys.argv[1]}] value {j['value']:.0f} sustained {j.get('sustained', {}).get('value')}  " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items()))
PY
done
done
