#!/bin/bash
# Same-box A/B of compile-time variants: builds a copy of the library per EXTRA flag set in /tmp and runs the short bench
# with each (GFO_LIB selects the library), printing value and the per-stage times.
# usage (through gpurun): tools/ab_variant.sh "<flags A>" "<flags B>" ...     ("" = the library as built)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
i=0
for FL in "$@"; do
  i=$((i+1))
  if [ -n "$FL" ]; then
    D=/tmp/gfo_var$i; rm -rf $D; mkdir -p $D/pkg; cp -r $R/include $D/include; cp -r $R/gf-orb-slam2_amd/csrc $D/pkg/csrc
    ( cd $D/pkg/csrc && rm -f *.o && make -s -j8 EXTRA="$FL" OUT=/tmp/libgfo_var$i.so ) || exit 1
    export GFO_LIB=/tmp/libgfo_var$i.so
  else
    unset GFO_LIB
  fi
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic ${GFO_AB_ARGS} > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { tail -5 gpurun_out/ab_$i.err; exit 1; }
    python - "$FL" gpurun_out/ab_$i.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = j["roofline"]["stage_ms_per_step"]
print(f"[{sys.argv[1] or 'as built'}] value {j['value']:.0f} sustained {j.get('sustained', {}).get('value')}  " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items()))
PY
  done
done
