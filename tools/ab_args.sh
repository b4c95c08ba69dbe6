#!/bin/bash
# Same-box A/B of bench arguments / environment: each argument is "ENV=V ENV2=V2 -- bench args" (either part may be empty)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
i=0
for SPEC in "$@"; do
  i=$((i+1))
  ENVS="${SPEC%%--*}"; ARGS="${SPEC#*--}"
  for rep in 1 2; do
    env $ENVS python bench.py --steps ${GFO_AB_STEPS:-100} --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic $ARGS > gpurun_out/aa_$i.json 2> gpurun_out/aa_$i.err || { tail -5 gpurun_out/aa_$i.err; exit 1; }
    python - "$SPEC" gpurun_out/aa_$i.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = j["roofline"]["stage_ms_per_step"]
print(f"[{sys.argv[1]}] value {j['value']:.0f} sustained {j.get('sustained', {}).get('value')}  " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items()))
PY
  done
done
