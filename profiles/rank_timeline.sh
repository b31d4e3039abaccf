#!/bin/bash
# Kernel timeline of ONE rank's job of an 8-way split (bench.py --tile 3/8, 16 steps = 256 frames = one launch): rocprofv3
# --kernel-trace of the bench command, then the last kernels of the process with their start / end relative to the timed
# raytrace launch -- where the ~2 ms a rank's job takes beyond its steady-state rate go (ramp, drain, the batched mean).
# usage: bash profiles/rank_timeline.sh <out-prefix under gpurun_out/> [bench args]
P=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
ARGS=${@:---tile 3/8 --steps 16 --warmup 2}
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/${P}_trace -- python3 $ROOT/bench.py --no-pmc --no-cpu-baseline --no-also --no-forest $ARGS > $OUT/${P}_bench.json 2> $OUT/${P}_trace.err)
python3 - "$(ls $OUT/${P}_trace/*/*_kernel_trace.csv | head -1)" $OUT/${P}_bench.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rt = [r for r in rows if "k_raytrace_sm" in r["Kernel_Name"]]
last = rt[-1]
t0 = last["s"]
print("bench line:", open(sys.argv[2]).read()[:160].strip())
print(f"timed raytrace launch: grid {last['Grid_Size_X']}, {(last['e'] - last['s']) / 1e3:.1f} us")
for r in rows:
    if r["s"] >= t0 - 3_000_000 and r["s"] <= last["e"] + 3_000_000:
        print(f"  {(r['s'] - t0) / 1e3:10.1f} .. {(r['e'] - t0) / 1e3:10.1f} us  ({(r['e'] - r['s']) / 1e3:9.1f} us)  {r['Kernel_Name'][:70]}")
PY
rm -rf $OUT/${P}_trace
