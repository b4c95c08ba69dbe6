#!/bin/bash
# Builds and runs the C latency harness on the GPU box: one EuRoC stereo pair through the C ABI, no Python.
# usage: tools/latency_pair.sh [tag]  -> gpurun_out/latency_pair_<tag>.json
R=${GRAFT_REPO_ROOT:-.}; TAG=${1:-x}; cd $R
gcc -O2 -I include tools/c/latency_pair.c -o /tmp/latency_pair -ldl -lpthread -lm || exit 1
/tmp/latency_pair gf-orb-slam2_amd/libgfo.so tests/golden 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency_pair_$TAG.json
