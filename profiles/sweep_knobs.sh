#!/bin/bash
# Knob sweeps for the state-machine kernels (1080p).  usage: sweep_knobs.sh [demo|dragon]
WL=${1:-demo}
run() { python bench.py --no-cpu-baseline --steps 64 --warmup 16 --workload $WL "$@" 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
echo "$WL in-order walk (variant 4): $(run --variant 4)"
for lm in 16 24 32 40 48; do echo "$WL deferred leaves, leaf_min $lm: $(MI3PT_LEAF_MIN=$lm run --variant 7)"; done
for wm in 24 40 48; do echo "$WL deferred leaves, leaf_min 32, walk_min $wm: $(MI3PT_WALK_MIN=$wm run --variant 7)"; done
