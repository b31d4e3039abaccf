#!/bin/bash
# FETCH_SIZE calibration (profiles/fetch_calibration.hip): the probe alone for its timings, then one rocprofv3 --pmc pass per
# counter group (never combined with trace domains other than --kernel-trace), then the table known bytes / counter.
# usage (on the GPU box): bash profiles/fetch_calibration.sh gpurun_out/fetchcal
set -u
OUT=${1:-gpurun_out/fetchcal}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/profiles/fetch_calibration.hip -o $ROOT/$OUT/fetch_calibration || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 $ROOT/$OUT/fetch_calibration > $ROOT/$OUT/timing.txt 2>&1 || { echo "probe failed rc=$?"; cat $ROOT/$OUT/timing.txt; exit 1; }
cat $ROOT/$OUT/timing.txt
rocprofv3 -L > $ROOT/$OUT/counters_available.txt 2>&1
i=0
while read -r counters; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $ROOT/$OUT/pass$i -- $ROOT/$OUT/fetch_calibration > $ROOT/$OUT/pass$i.log 2>&1 || echo "pass $i ($counters) rc=$?"
done <<'LIST'
FETCH_SIZE
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum
TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum
LIST
python3 $ROOT/profiles/fetch_calibration_summary.py $ROOT/$OUT
