#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command itself (default arguments and the driver's), with the line that
# process printed: gpurun_out/<tag>_dragon_1080p_kernel_{stats,trace}_<args>.csv, <tag>_bench_under_rocprof_<args>.json.
# usage: bash profiles/rocprof_stats.sh <tag>
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for A in "" "--steps 20 --warmup 5"; do
  S=$(echo "$A" | tr -d ' -' ); S=${S:-default}
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_$S -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --no-forest $A > $OUT/${TAG}_bench_under_rocprof_$S.json 2> $OUT/${TAG}_stats_$S.err)
  cp $(ls $OUT/${TAG}_stats_$S/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_dragon_1080p_kernel_stats_$S.csv
  cp $(ls $OUT/${TAG}_stats_$S/*/*_kernel_trace.csv | head -1) $OUT/${TAG}_dragon_1080p_kernel_trace_$S.csv
  rm -rf $OUT/${TAG}_stats_$S
done
echo "rocprof stats done"
