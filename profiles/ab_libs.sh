#!/bin/bash
# A/B of two builds of the library on ONE box: webgpu-pathtracer_amd/libA.so and libB.so (copied over libmi3pt.so in turn),
# alternating, the driver's bench arguments.  usage: bash profiles/ab_libs.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for r in $(seq 1 ${1:-3}); do
  for L in ${LIBS:-A B}; do
    cp webgpu-pathtracer_amd/lib$L.so webgpu-pathtracer_amd/libmi3pt.so; touch webgpu-pathtracer_amd/libmi3pt.so
    python bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$L: dragon %.0f  demo %.0f' % (j['value'], j['also']['demo']['value']))"
  done
done
