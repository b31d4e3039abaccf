#!/bin/bash
# One bench line (dragon + demo, no counter passes, no CPU baseline) per knob setting.
# usage: bash profiles/knob_sweep2.sh "<ENV=val ...>" "<ENV=val ...>" ...      -> gpurun_out/knob_sweep2.log
# (needs the EXPERIMENT build of the library, which maps MI3PT_<NAME> variables onto mi3pt_debug_set_option:
#  make -C webgpu-pathtracer_amd/csrc experiments; the release library reads no such variable)
export MI3PT_LIBRARY=${MI3PT_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/webgpu-pathtracer_amd/libmi3pt_exp.so}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
: > gpurun_out/knob_sweep2.log
for K in "$@"; do
  L=$(env $K python bench.py --no-pmc --no-cpu-baseline --steps 12 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('dragon %.0f  demo %.0f  ms_per_step %.3f kernel_ms %.3f' % (j['value'], j['also']['demo']['value'], j['ms_per_step'], j['roofline']['kernel_ms']))")
  echo "$K: $L" | tee -a gpurun_out/knob_sweep2.log
done
