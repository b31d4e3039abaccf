#!/bin/bash
# Collects PMC counters for the dominant kernel in separate passes (never combined with
# trace domains other than --kernel-trace).  STEPS / WARMUP (env) set the bench length, PASSES
# (env, e.g. "4 5") restricts the counter groups.  TA_* and GRBM_* passes are left out: on this
# pool a TA_* pass aborted rocprofv3 (signal 6) and the following pass hung, and a TCP_* pass
# (TCP_TOTAL_CACHE_ACCESSES_sum, TCP_TCC_READ_REQ_sum, TCP_GATE_EN1/2_sum) hung the run until its timeout (round 1).  usage: pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  i=$((i+1))
  if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $ROOT/$OUT/pass$i -- python3 $ROOT/bench.py --steps ${STEPS:-4} --warmup ${WARMUP:-1} --no-cpu-baseline "$@" > $ROOT/$OUT.pass$i.log 2>&1
done <<'LIST'
SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
LIST
