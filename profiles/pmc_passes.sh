#!/bin/bash
# Collects PMC counters for the dominant kernel in separate passes (never combined with trace
# domains other than --kernel-trace), with the same job bench.py times: `bench.py --inner-pmc`
# (one launch at a time; the library switches its launch gate off by itself when a profiler is attached --
# mi3pt_create: counter collection serialises kernels in interception order, which deadlocked gated,
# overlapping launches intermittently).
# bench.py runs these three passes itself after its timed job (roofline.pmc_counters); this script
# keeps the raw per-dispatch CSVs for cross-checking.  TA_* / TCP_* / GRBM_* passes are left out:
# on this pool they aborted or hung rocprofv3 (round 1).
# usage: pmc_passes.sh <outdir> [bench args, e.g. --steps 20 --warmup 5]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $ROOT/$OUT/pass$i -- python3 $ROOT/bench.py --inner-pmc "$@" > $ROOT/$OUT.pass$i.log 2>&1 || echo "pass $i ($counters) rc=$?"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum
LIST
