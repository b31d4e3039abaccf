#!/bin/bash
# Knob sweeps for the state-machine kernel (demo workload, 1080p).  usage: sweep_knobs.sh
run() { python bench.py --no-cpu-baseline --steps 32 "$@" 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
for w in 8 10 12 16; do echo "waves/cu $w pipelined: $(MI3PT_WAVES_PER_CU=$w run)"; done
for w in 10 16; do echo "waves/cu $w unpipelined: $(MI3PT_PIPELINE=0 MI3PT_WAVES_PER_CU=$w run)"; done
for wm in 24 32 40 48; do echo "walk_min $wm (16 waves/cu): $(MI3PT_WALK_MIN=$wm run)"; done
