#!/bin/bash
# Kernel + copy timeline of the END of any python tool (through gpurun): tools/trace_cmd_tail.sh <tag> <n events> <script.py> [args]
cd /tmp && export TMPDIR=/tmp
TAG=$1; N=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_tail_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys
root, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]) for r in csv.DictReader(open(f[0]))] if f else []
c = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)
if c: rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[-16:]) for r in csv.DictReader(open(c[0]))]
rows.sort()
rows = rows[-n:]
t0 = rows[0][0] if rows else 0
for s, e, k in rows: print(f"{(s-t0)/1e3:10.1f} +{(e-s)/1e3:8.1f} {k}")
PY
