#!/bin/bash
# Builds and runs the K-stream per-frame harness on the GPU box (tools/c/boundary_throughput.c).
# usage: tools/boundary_throughput.sh [tag] [seconds] [modes] [K list] [combine 0|1] [pin 0|1]  -> gpurun_out/boundary_throughput_<tag>.json
R=${GRAFT_REPO_ROOT:-.}; TAG=${1:-x}; SEC=${2:-2}; MODES=${3:-both}; KL=${4:-1,2,4,8,16}; CMB=${5:-1}; PIN=${6:-0}; cd $R
mkdir -p gpurun_out
gcc -O2 -I include tools/c/boundary_throughput.c -o /tmp/boundary_throughput -ldl -lpthread -lm || exit 1
timeout -k 10 300 /tmp/boundary_throughput gf-orb-slam2_amd/libgfo.so tests/golden $SEC $MODES $KL $CMB $PIN 2> gpurun_out/boundary_throughput_$TAG.err | tee gpurun_out/boundary_throughput_$TAG.json
rc=$?
grep -v amdgpu.ids gpurun_out/boundary_throughput_$TAG.err | tail -5
exit $rc
