#!/bin/bash
# cull_probe (box / triangle tests per ray, ms per frame, image identity) for several builds of the library, each named by
# MI3PT_LIBRARY (the Python host reads it): the built libmi3pt.so is not touched.
# usage: LIBS="_ab/libA.so _ab/libB.so" bash profiles/ab_cull_probe.sh [scene ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for S in ${@:-demo dragon}; do
  for L in ${LIBS:-webgpu-pathtracer_amd/libmi3pt.so}; do
    echo "== $L $S"
    if [ "$S" = forest ]; then MI3PT_LIBRARY=$ROOT/$L python profiles/cull_probe.py forest 3840x2160 8; else MI3PT_LIBRARY=$ROOT/$L python profiles/cull_probe.py $S 1920x1080 32; fi
  done
done
