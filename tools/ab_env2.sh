#!/bin/bash
# Same-box A/B of run-time settings: tools/ab_env2.sh "VAR=a" "VAR=b VAR2=c" ...  ("" = defaults); two runs each, interleaved
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for rep in 1 2; do
for E in "$@"; do
  env $E python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic --no-verify > gpurun_out/abe.json 2> gpurun_out/abe.err || { tail -5 gpurun_out/abe.err; exit 1; }
  python3 - "$E" <<'PY'
import json, sys
j = json.loads(open('gpurun_out/abe.json').read().strip().splitlines()[-1])
st = j["roofline"]["stage_ms_per_step"]
print(f"[{sys.argv[1] or 'defaults'}] value {j['value']:.0f} sustained {j.get('sustained', {}).get('value')}  " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items()))
PY
done
done
