#!/bin/bash
# Collects the measurement artefacts of a round on the GPU box into gpurun_out/<tag>_*: the bench
# lines (default arguments and the driver's --steps 20 --warmup 5; demo workload too), rocprofv3
# kernel stats + kernel trace of the bench command itself, the scaling model, the culling probe
# on BASELINE configs 2 / 3 / 5, wave timelines, the Node render-loop bench, fullscreen timing.
# bench.py runs its own PMC passes (roofline.traffic, valu_issue_frac, ...), so none are run here.
# usage: bash profiles/collect_round.sh <tag>      (about 10 GPU-minutes)
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
# Order matters on this pool: a box that has just run counter (--pmc) passes is slow for a while (clocks), so everything
# that is timed without counters runs first, and the three bench lines -- each ends with its own counter passes -- are
# spaced out.
if [ -z "${SKIP_SCALING:-}" ]; then      # (every rank of every split since round 5: ~3 minutes each; SKIP_SCALING=1 when it ran in a call of its own)
python profiles/scaling_model.py --steps 20 --warmup 5 > $OUT/${TAG}_scaling_model_steps20.log 2>&1
python profiles/scaling_model.py --steps 16 --warmup 2 > $OUT/${TAG}_scaling_model_steps16.log 2>&1
python profiles/scaling_model.py --steps 4 --warmup 1 > $OUT/${TAG}_scaling_model_steps4.log 2>&1
echo "scaling model done"
fi
(python profiles/cull_probe.py demo; python profiles/cull_probe.py dragon; python profiles/cull_probe.py forest 3840x2160 8) > $OUT/${TAG}_cull_probe.log 2>&1
echo "cull probe done"
# wave timelines: the LEAN kernel with lane counts (experiment build: what ships + a dozen scalar counters, five waves) and the
# four-wave diagnostic twin (clock stamps per step: cycle split and drain statistics; half the shipped kernel's speed)
if [ -f webgpu-pathtracer_amd/libmi3pt_exp.so ]; then
  for W in demo dragon forest; do
    MI3PT_LIBRARY=$ROOT/webgpu-pathtracer_amd/libmi3pt_exp.so LITE=1 WORKLOAD=$W python profiles/wave_timeline.py 1920x1080 64 > $OUT/${TAG}_wave_timeline_${W}_lite.log 2>&1
  done
fi
WORKLOAD=dragon python profiles/wave_timeline.py 1920x1080 64 > $OUT/${TAG}_wave_timeline_dragon_twin.log 2>&1
WORKLOAD=dragon TILE=3/8 python profiles/wave_timeline.py 1920x1080 256 > $OUT/${TAG}_wave_timeline_dragon_rank3of8_256frames_twin.log 2>&1
bash profiles/rank_timeline.sh ${TAG}_rank3of8 > $OUT/${TAG}_rank_timeline.log 2>&1
python profiles/probe_spf.py dragon frames > $OUT/${TAG}_spf_same_frames.log 2>&1
(for W in demo dragon; do echo "== $W"; WORKLOAD=$W python profiles/probe_interactive.py 64; done) > $OUT/${TAG}_interactive.log 2>&1
python profiles/fullscreen_time.py > $OUT/${TAG}_fullscreen_time.log 2>&1
python -c "
import sys; sys.path.insert(0, 'webgpu-pathtracer_amd/py')
from mi3pt_host import scenes
scenes.synthetic_env().tofile('/tmp/env.f32')"
node webgpu-pathtracer_amd/js/tools/bench_render_loop.js --env /tmp/env.f32 --frames 64 > $OUT/${TAG}_node_render_loop.json 2>&1
echo "timelines + loop bench done"
bash profiles/rocprof_stats.sh $TAG
python bench.py --no-forest > $OUT/${TAG}_dragon_1080p_bench.json 2> $OUT/${TAG}_dragon_bench.err
echo "bench (default args) done: $(cut -c1-120 $OUT/${TAG}_dragon_1080p_bench.json)"
sleep 30
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_dragon_1080p_bench_driver_args.json 2>/dev/null
echo "bench (driver args) done: $(cut -c1-120 $OUT/${TAG}_dragon_1080p_bench_driver_args.json)"
sleep 30
python bench.py --workload demo --no-forest > $OUT/${TAG}_demo_1080p_bench.json 2>/dev/null
echo "bench (demo) done: $(cut -c1-120 $OUT/${TAG}_demo_1080p_bench.json)"
STEPS=20; WARM=5
bash profiles/pmc_passes.sh gpurun_out/${TAG}_pmc_dragon --steps $STEPS --warmup $WARM
python profiles/pmc_summary.py $OUT/${TAG}_pmc_dragon k_raytrace $(( (STEPS * 16 + 255) / 256 )) > $OUT/${TAG}_dragon_pmc_per_launch.txt
rm -rf $OUT/${TAG}_pmc_dragon $OUT/${TAG}_pmc_dragon.pass*.log
echo "pmc passes done: $(grep -c . $OUT/${TAG}_dragon_pmc_per_launch.txt) counters"
# config 5's scene (the one larger than the Infinity Cache): the forest leg of bench.py, 1920x1080, 8 steps = one 128-frame launch
bash profiles/pmc_passes.sh gpurun_out/${TAG}_pmc_forest --workload forest --steps 8 --warmup 4
python profiles/pmc_summary.py $OUT/${TAG}_pmc_forest k_raytrace 1 > $OUT/${TAG}_forest_pmc_per_launch.txt
rm -rf $OUT/${TAG}_pmc_forest $OUT/${TAG}_pmc_forest.pass*.log
echo "forest pmc passes done: $(grep -c . $OUT/${TAG}_forest_pmc_per_launch.txt) counters"
