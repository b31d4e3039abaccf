#!/bin/bash
# cull_probe (box / triangle tests per ray, ms per frame, image identity) for several builds of the library:
# webgpu-pathtracer_amd/lib$L.so copied over libmi3pt.so in turn.  usage: LIBS="A B" bash profiles/ab_cull_probe.sh [scene ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for S in ${@:-demo dragon}; do
  for L in ${LIBS:-A B}; do
    cp webgpu-pathtracer_amd/lib$L.so webgpu-pathtracer_amd/libmi3pt.so; touch webgpu-pathtracer_amd/libmi3pt.so
    echo "== lib$L $S"
    if [ "$S" = forest ]; then python profiles/cull_probe.py forest 3840x2160 8; else python profiles/cull_probe.py $S 1920x1080 32; fi
  done
done
