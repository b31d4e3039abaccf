#!/bin/bash
# One bench line of ONE rank of an N-way tile split (bench.py --tile R/N, the driver's arguments) per knob setting.
# usage: TILE=0/8 bash profiles/knob_sweep3.sh "<ENV=val ...>" ...      -> gpurun_out/knob_sweep3.log
# (needs the EXPERIMENT build of the library, which maps MI3PT_<NAME> variables onto mi3pt_debug_set_option:
#  make -C webgpu-pathtracer_amd/csrc experiments; the release library reads no such variable)
export MI3PT_LIBRARY=${MI3PT_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/webgpu-pathtracer_amd/libmi3pt_exp.so}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
TILE=${TILE:-0/8}
for K in "$@"; do
  L=$(env $K python bench.py --no-pmc --no-cpu-baseline --no-also --steps 20 --warmup 5 --tile $TILE 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('%.3f ms for the job, frames/launch %s kernel_ms %.3f' % (j['ms_per_step'] * j['steps'], j['roofline']['frames_per_launch'], j['roofline']['kernel_ms']))")
  echo "tile $TILE $K: $L" | tee -a gpurun_out/knob_sweep3.log
done
