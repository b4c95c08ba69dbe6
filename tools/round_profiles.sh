#!/bin/bash
# The measurement records of a round in one GPU call (through gpurun): kernel trace + PMC traffic of the bench command,
# SQ counters, the per-frame boundary harness (pageable / pinned caller buffers / combining off), per-frame latency of the
# three call patterns, the K = 8 kernel trace of the boundary harness with the frame combiner, and the default bench line.
# usage: tools/round_profiles.sh <tag>     -> gpurun_out/*_<tag>*  (copy what is to be kept into profiles/)
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tools/profile_gpu.sh $TAG > gpurun_out/profile_$TAG.log 2>&1 || { tail -5 gpurun_out/profile_$TAG.log; exit 1; }
tools/pmc_sq.sh > gpurun_out/sq_counters_$TAG.txt 2>&1 || { tail -5 gpurun_out/sq_counters_$TAG.txt; exit 2; }
tools/boundary_throughput.sh ${TAG}_pageable 3 both 1,2,4,8,16 1 0 > /dev/null || exit 3
tools/boundary_throughput.sh ${TAG}_pinned 3 stereo 1,2,4,8,16 1 1 > /dev/null || exit 4
tools/boundary_throughput.sh ${TAG}_nocombine 3 both 1,2,4,8,16 0 0 > /dev/null || exit 5
tools/latency_pair.sh $TAG > /dev/null || exit 6
tools/trace_boundary.sh ${TAG}_k8_combined 8 stereo 0 1 > /dev/null || exit 8
python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err || exit 9
echo "round profiles $TAG done"
