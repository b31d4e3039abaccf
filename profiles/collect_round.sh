#!/bin/bash
# Collects the measurement artefacts of a round on the GPU box into gpurun_out/<tag>_*:
# bench lines (demo with CPU baseline, dragon), rocprofv3 kernel stats of the bench command,
# PMC passes (instruction mix, HBM traffic) for both workloads, wave timelines, fullscreen
# timing.  usage: bash profiles/collect_round.sh <tag>      (about 5 GPU-minutes)
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${TAG}_demo_1080p_bench.json 2> $OUT/${TAG}_demo_bench.err
echo "demo bench done: $(cut -c1-100 $OUT/${TAG}_demo_1080p_bench.json)"
python bench.py --no-cpu-baseline --workload dragon > $OUT/${TAG}_dragon_1080p_bench.json 2>/dev/null
echo "dragon bench done: $(cut -c1-100 $OUT/${TAG}_dragon_1080p_bench.json)"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py > $OUT/${TAG}_stats.log 2>&1)
cp $(ls $OUT/${TAG}_stats/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_demo_1080p_kernel_stats.csv
echo "rocprof stats done"
PASSES="1 2 4 5" STEPS=64 WARMUP=16 bash profiles/pmc_passes.sh gpurun_out/${TAG}_pmc_demo
python profiles/pmc_summary.py $OUT/${TAG}_pmc_demo > $OUT/${TAG}_demo_pmc_per_16frame_launch.txt
echo "demo pmc done"
PASSES="1 4 5" STEPS=64 WARMUP=16 bash profiles/pmc_passes.sh gpurun_out/${TAG}_pmc_dragon --workload dragon
python profiles/pmc_summary.py $OUT/${TAG}_pmc_dragon > $OUT/${TAG}_dragon_pmc_per_16frame_launch.txt
echo "dragon pmc done"
python profiles/wave_timeline.py 1920x1080 16 > $OUT/${TAG}_wave_timeline_demo.log 2>&1
WORKLOAD=dragon python profiles/wave_timeline.py 1920x1080 16 > $OUT/${TAG}_wave_timeline_dragon.log 2>&1
python profiles/fullscreen_time.py > $OUT/${TAG}_fullscreen_time.log 2>&1
echo "timelines done"
