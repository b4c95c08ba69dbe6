#!/bin/bash
# Duration of k_fast truncated after each phase (instrumented build, GFO_FAST_STOP: 1 = tile load, 2 = + stage A, 3 = + stage B,
# 4 = first round only, 0 = whole kernel) and its instruction counts; results of the truncated runs are wrong by construction.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
D=/tmp/gfo_dbg; rm -rf $D; mkdir -p $D/pkg; cp -r $R/include $D/include; cp -r $R/gf-orb-slam2_amd/csrc $D/pkg/csrc
( cd $D/pkg/csrc && rm -f *.o && make -s -j8 EXTRA=-DGFO_FAST_DEBUG OUT=/tmp/libgfo_dbg.so ) || exit 1
export GFO_LIB=/tmp/libgfo_dbg.so
for s in 1 2 3 4 0; do
  GFO_FAST_STOP=$s python bench.py --steps 40 --warmup 10 --streams 1 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic > gpurun_out/fpt.json 2> gpurun_out/fpt.err || { tail -3 gpurun_out/fpt.err; exit 1; }
  python - $s gpurun_out/fpt.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"stop {sys.argv[1]}: k_fast {j['roofline']['stage_ms_per_step']['fast']*1e3:.1f} us per {j['config']['images_per_step_per_gpu']} images")
PY
done
