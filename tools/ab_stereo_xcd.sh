#!/bin/bash
# k_stereo_match_rows with all bands of a pair on one XCD (GFO_STEREO_XCD=1, the default) against the plain (bands, pairs) grid:
# FETCH_SIZE / WRITE_SIZE per launch (two PMC passes each) and the stage times of the short bench, on ONE box.
# usage (through gpurun): tools/ab_stereo_xcd.sh   -> gpurun_out/ab_stereo_xcd.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; OUTF=$R/gpurun_out/ab_stereo_xcd.txt; : > $OUTF
for X in 0 1; do
  export GFO_STEREO_XCD=$X
  for CTR in FETCH_SIZE WRITE_SIZE; do
    OUT=/tmp/ab_sx_${X}_$CTR; rm -rf $OUT
    timeout -k 10 200 rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 bench.py --pmc-child --steps 6 --warmup 2 --streams 1 > $OUT.log 2>&1 || { tail -3 $OUT.log; exit 1; }
    python3 - $OUT $CTR $X <<'PY' | tee -a $OUTF
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)[0]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if row['Counter_Name'] == sys.argv[2]: acc[row['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(row['Counter_Value']))
print(f"GFO_STEREO_XCD={sys.argv[3]} {sys.argv[2]} MB per launch: " + "  ".join(f"{k}={sum(v)/len(v)/1024:.1f}" for k, v in acc.items() if k.startswith('k_stereo')))
PY
  done
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-other-configs --no-boundary --no-live-traffic --no-verify > gpurun_out/absx.json 2> gpurun_out/absx.err || { tail -5 gpurun_out/absx.err; exit 1; }
    python3 - $X <<'PY' | tee -a $OUTF
import json, sys
j = json.loads(open('gpurun_out/absx.json').read().strip().splitlines()[-1])
st = j["roofline"]["stage_ms_per_step"]
print(f"GFO_STEREO_XCD={sys.argv[1]} value {j['value']:.0f}  " + " ".join(f"{k}={v*1e3:.1f}" for k, v in st.items() if k.startswith('stereo')))
PY
  done
done
