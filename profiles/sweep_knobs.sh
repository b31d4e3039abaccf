#!/bin/bash
# Knob sweeps for the state-machine kernels (1080p).  usage: sweep_knobs.sh [demo|dragon]
# (needs the EXPERIMENT build of the library, which maps MI3PT_<NAME> variables onto mi3pt_debug_set_option:
#  make -C webgpu-pathtracer_amd/csrc experiments; the release library reads no such variable)
export MI3PT_LIBRARY=${MI3PT_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/webgpu-pathtracer_amd/libmi3pt_exp.so}
WL=${1:-demo}
run() { python bench.py --no-cpu-baseline --steps ${STEPS:-64} --warmup 16 --workload $WL "$@" 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
echo "$WL default: $(run)"
echo "$WL in-order walk (variant 4): $(run --variant 4)"
for lm in 24 40; do echo "$WL leaf_min $lm: $(MI3PT_LEAF_MIN=$lm run)"; done
for wm in 24 28 36 40; do echo "$WL walk_min $wm: $(MI3PT_WALK_MIN=$wm run)"; done
for b in 8 32; do echo "$WL batch $b: $(MI3PT_BATCH=$b STEPS=128 run)"; done
echo "$WL batch 16, 128 steps: $(STEPS=128 run)"
