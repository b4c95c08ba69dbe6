#!/bin/bash
# SQ-side counters per kernel (one pass, 8 SQ slots): where the waves spend their cycles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --profile-steps 1 > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --profile-steps 1 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_sq')
for sub in ('a','b'):
    f=glob.glob(os.path.join(root,sub,'**','*counter_collection.csv'),recursive=True)
    if not f: print('no csv',sub); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k,v in acc.items():
        if not k.startswith('k_'): continue
        line = k + ' ' + ' '.join(f"{c}={sum(x)/len(x):.4g}" for c,x in sorted(v.items()))
        print(line)
        open(os.path.join(root, 'summary.txt'), 'a').write(line + '\n')
# per stage and STEP (a step = one k_fast launch; the pyramid is several launches per step): what bench.py's issue_frac reads
import json, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from bench import STAGE_OF_KERNEL
per = collections.defaultdict(lambda: collections.defaultdict(float)); steps = collections.Counter()
for sub in ('a','b'):
    f=glob.glob(os.path.join(root,sub,'**','*counter_collection.csv'),recursive=True)
    if not f: continue
    for row in csv.DictReader(open(f[0])):
        kn=row['Kernel_Name']
        for pat, st in STAGE_OF_KERNEL:
            if pat in kn and not (pat == 'k_stereo_match' and 'sad' in kn):
                per[st][row['Counter_Name']] += float(row['Counter_Value'])
                if st == 'fast': steps[row['Counter_Name']] += 1
                break
out = {st: {c: v / max(1, steps.get(c, 0) or max(steps.values())) for c, v in cs.items()} for st, cs in per.items()}
json.dump({"workload": "stereo752", "batch": 256, "per_stage_per_step": out,
           "note": "rocprofv3 --pmc SQ_* passes of `bench.py --steps 3 --warmup 1` (tools/pmc_sq.sh), summed per stage and divided by the steps seen"},
          open(os.path.join(root, 'sq_per_step.json'), 'w'), indent=1)
PY
