// the product file, unmodified, for tests/host/gfo_api_san.cc (fake HIP runtime: tests/host/fakehip)
#include "../../gf-orb-slam2_amd/csrc/gfo_combine.hip"
