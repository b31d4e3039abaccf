#!/bin/bash
# usage: bash profiles/gate_trace.sh <outdir>
OUT=${1:-gpurun_out/gate_trace}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $ROOT/$OUT
cd $ROOT
for S in -1 0 1 -1; do
  (export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/t -- python3 $ROOT/profiles/probe_gate_trace.py $S) > $ROOT/$OUT/run_$S.log 2>&1
  echo "== six_waves option $S: $(grep gate $ROOT/$OUT/run_$S.log)"
  python3 - <<PY
import csv,glob
f=glob.glob("$ROOT/$OUT/t/*/*_kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f))); t0=min(int(r['Start_Timestamp']) for r in rows)
for r in sorted(rows,key=lambda r:int(r['Start_Timestamp'])):
    n=r['Kernel_Name']
    if 'k_raytrace_sm' in n: print(f"   {(int(r['Start_Timestamp'])-t0)/1e6:10.3f} .. {(int(r['End_Timestamp'])-t0)/1e6:10.3f} ms ({(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:7.3f})  ...{n[-22:]}")
PY
  rm -rf $ROOT/$OUT/t
done
echo "== under rocprofv3 --pmc SQ_WAVES (counter collection): the gate must be off"
(export TMPDIR=/tmp; cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $ROOT/$OUT/p -- python3 $ROOT/profiles/probe_gate_trace.py -1) > $ROOT/$OUT/run_pmc.log 2>&1
grep gate $ROOT/$OUT/run_pmc.log; rm -rf $ROOT/$OUT/p
