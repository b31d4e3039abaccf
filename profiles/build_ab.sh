#!/bin/bash
# Builds a named variant of the library for an A/B run without touching libmi3pt.so:  _ab/lib<NAME>.so from the working tree's
# sources with extra compiler flags.  usage: bash profiles/build_ab.sh NAME [-DPT_X_...=... ...]     (then profiles/ab_quick.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
mkdir -p $ROOT/_ab
cd $ROOT/webgpu-pathtracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function "$@" -shared -o $ROOT/_ab/lib$NAME.so \
  pt_kernels.hip pt_context.hip pt_lbvh.hip pt_host_scene.cpp pt_host_wide.cpp -lpthread && echo "built _ab/lib$NAME.so ($*)"
