#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Outputs land under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/.
set -o pipefail
TAG=${1:-r01}
ARGS=${2:---steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --pmc-child --steps 6 --warmup 2 --streams 1 > $OUT/bench_fetch.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --pmc-child --steps 6 --warmup 2 --streams 1 > $OUT/bench_write.log 2>&1 || exit 3
find $OUT -name "*.csv" | head -20
python3 tools/summarize_profile.py $OUT $TAG || exit 4
