#!/bin/bash
# same-box A/B of two k_fast.hip versions: builds an alternate libgfo with tools/ab/k_fast_old.hip
R=$GRAFT_REPO_ROOT; cd $R
D=/tmp/gfo_ab; rm -rf $D; mkdir -p $D/pkg; cp -r include $D/include; cp -r gf-orb-slam2_amd/csrc $D/pkg/csrc
cp tools/ab/k_fast_old.hip $D/pkg/csrc/k_fast.hip
( cd $D/pkg/csrc && rm -f *.o && make -s OUT=/tmp/libgfo_old.so ) || exit 1
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export GFO_LIB=/tmp/libgfo_old.so; else unset GFO_LIB; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --streams 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['roofline']['stage_ms_per_step']['fast'])"
  done
done
