#!/bin/bash
# Same-box A/B of bench.py's context count and chain stage (CONTEXTS / CHAIN_STAGE in bench.py): usage (through gpurun): tools/ab_contexts.sh
# round 4: 2 chained after the pyramid 279.3-280.2 k (stable), 2 free-running 277.6-283.7 k (bimodal), 3 contexts 273-275 k, chained after FAST 275-277 k
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for A in "--streams 2" "--streams 3" "--streams 2:GFO_BENCH_CHAIN=2" "--streams 2:GFO_BENCH_CHAIN=0"; do
  ARGS=${A%%:*}; ENVV=${A#*:}; [ "$ENVV" = "$A" ] && ENVV="X=1"
  env $ENVV python bench.py $ARGS --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic --no-verify 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$A]', round(j['value']), j.get('sustained',{}).get('value'))"
done
done
