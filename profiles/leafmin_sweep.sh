#!/bin/bash
# (needs the EXPERIMENT build of the library, which maps MI3PT_<NAME> variables onto mi3pt_debug_set_option:
#  make -C webgpu-pathtracer_amd/csrc experiments; the release library reads no such variable)
export MI3PT_LIBRARY=${MI3PT_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/webgpu-pathtracer_amd/libmi3pt_exp.so}
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --no-pmc --no-cpu-baseline --steps 8 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"], j["also"]["demo"]["value"])'; }
for lm in 16 24 32 40; do echo "leaf_min $lm: $(MI3PT_LEAF_MIN=$lm run)"; done
