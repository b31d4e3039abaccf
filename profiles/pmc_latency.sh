#!/bin/bash
# Where are the L2 misses served from, and how long do they take?  TCC_EA0_RDREQ_LEVEL (fabric read requests in flight, summed per cycle) over
# TCC_EA0_RDREQ (requests) = mean cycles a request is outstanding -- for the calibration probe's patterns (a 4 GiB table: HBM; a 128 MiB table:
# Infinity Cache) and for the raytrace kernel on the dragon-class scene (181 MB) and the forest (2 GB).
# usage (GPU box): bash profiles/pmc_latency.sh gpurun_out/lat
set -u
OUT=${1:-gpurun_out/lat}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/profiles/fetch_calibration.hip -o $ROOT/$OUT/fetch_calibration || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 $ROOT/$OUT/fetch_calibration > $ROOT/$OUT/timing.txt 2>&1 || exit 1
C="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_CYCLE_sum"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/pass1 -- $ROOT/$OUT/fetch_calibration > $ROOT/$OUT/pass1.log 2>&1 || echo "probe pass rc=$?"
python3 $ROOT/profiles/fetch_calibration_summary.py $ROOT/$OUT > $ROOT/$OUT/probe_latency.txt
rm -f $ROOT/$OUT/fetch_calibration
for W in dragon forest; do
  A="--steps 20 --warmup 5"; [ $W = forest ] && A="--steps 8 --warmup 4"
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/${W}/pass1 -- python3 $ROOT/bench.py --inner-pmc --workload $W $A > $ROOT/$OUT/$W.log 2>&1 || echo "$W pass rc=$?"
  python3 $ROOT/profiles/pmc_summary.py $ROOT/$OUT/$W k_raytrace > $ROOT/$OUT/${W}_latency.txt
done
rm -rf $ROOT/$OUT/pass1 $ROOT/$OUT/dragon $ROOT/$OUT/forest
tail -n +1 $ROOT/$OUT/probe_latency.txt $ROOT/$OUT/dragon_latency.txt $ROOT/$OUT/forest_latency.txt
