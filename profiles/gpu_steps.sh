#!/bin/bash
# Runs the given shell steps one after the other on the GPU box, each under its own timeout, each
# with its output in gpurun_out/<tag>_<n>.log.  An ordinary failure (a failing test) does not stop
# the list; a step that had to be killed (timeout) does -- nothing else is started on a GPU that
# may be wedged.  usage: gpu_steps.sh <tag> <seconds-per-step> "<cmd 1>" "<cmd 2>" ...
TAG=$1; LIMIT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
n=0
for cmd in "$@"; do
  n=$((n+1))
  log=gpurun_out/${TAG}_${n}.log
  echo "== step $n: $cmd" | tee $log
  timeout -k 10 $LIMIT bash -c "$cmd" >> $log 2>&1
  rc=$?
  echo "== step $n rc=$rc: $(tail -n 3 $log | tr '\n' ' ' | cut -c1-300)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n was killed at its limit: stopping"; exit 1; fi
done
exit 0
