#!/bin/bash
# Same-box A/B of bench arguments + environment: tools/ab_args2.sh "ENV=.. -- --args" ...   (part before " -- " is the environment)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for rep in 1 2; do
for S in "$@"; do
  E="${S%% -- *}"; A="${S#* -- }"; [ "$S" == "$E" ] && { E=""; A="$S"; }
  env $E python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic $A > gpurun_out/aba.json 2> gpurun_out/aba.err || { tail -5 gpurun_out/aba.err; exit 1; }
  python - "$S" gpurun_out/aba.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]}] value {j['value']:.0f} sustained {j.get('sustained', {}).get('value')} ms/step {j['ms_per_step']:.4f}")
PY
done
done
