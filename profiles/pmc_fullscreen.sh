#!/bin/bash
# SQ counters of k_fullscreen (de-noise on, 1920x1080): two passes over profiles/fullscreen_time.py, per-launch means of
# the launches with de-noise (the longest ones).  usage: bash profiles/pmc_fullscreen.sh <tag>
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/${TAG}_fs_pmc/pass$i -- python3 $ROOT/profiles/fullscreen_time.py 1920x1080 > $OUT/${TAG}_fs_pmc.pass$i.log 2>&1 || echo "pass $i rc=$?"
done <<'LIST'
SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
LIST
cd $ROOT
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/${TAG}_fs_pmc/pass*/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(p)) if "k_fullscreenE" in r["Kernel_Name"] or r["Kernel_Name"].startswith("pt::k_fullscreen(")]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, v in by.items():
        v = sorted(v)[-6:]          # the de-noise launches at scaling 1 do the most of everything
        print(f"{name:28s} {sum(v) / len(v):16.0f}   (mean of the {len(v)} largest of {len(by[name])} launches)")
PY
rm -rf $OUT/${TAG}_fs_pmc $OUT/${TAG}_fs_pmc.pass*.log
