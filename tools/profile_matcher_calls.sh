#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/matcher_call_latency.py (through gpurun): per-kernel calls / average duration of the
# host-array matcher calls -> gpurun_out/kernel_stats_matcher_calls_<tag>.csv
cd /tmp && export TMPDIR=/tmp
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_matcher_calls_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/matcher_call_latency.py > $OUT/log.txt 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] || { tail -5 $OUT/log.txt; exit 1; }
{ echo "# rocprofv3 --kernel-trace --stats -- python3 tools/matcher_call_latency.py (EuRoC frame, 2008 keypoints; every call 55 times)"; cat $F; } > $R/gpurun_out/kernel_stats_matcher_calls_$TAG.csv
grep -v "amdgpu.ids\|rocprofv3\|HSA ver\|Opened result" $OUT/log.txt | tail -8
head -16 $R/gpurun_out/kernel_stats_matcher_calls_$TAG.csv | cut -c1-150
