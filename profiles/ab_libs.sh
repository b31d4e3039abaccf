#!/bin/bash
# A/B of builds of the library on ONE box, alternating, the driver's bench arguments; each build is named by MI3PT_LIBRARY (read by
# the Python host): the built libmi3pt.so is not touched.  (profiles/ab_quick.py is the faster tool: scenes generated once.)
# usage: LIBS="_ab/libA.so _ab/libB.so" bash profiles/ab_libs.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for r in $(seq 1 ${1:-3}); do
  for L in ${LIBS:-webgpu-pathtracer_amd/libmi3pt.so}; do
    MI3PT_LIBRARY=$ROOT/$L python bench.py --no-pmc --no-cpu-baseline --no-forest --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$L: dragon %.0f  demo %.0f' % (j['value'], j['also']['demo']['value']))"
  done
done
