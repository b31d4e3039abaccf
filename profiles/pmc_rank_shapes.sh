#!/bin/bash
# Counter passes for the shapes an N-GPU bench line needs (round-4 verdict, next #2a): every rank listed below of an N-way tile
# split rendered ALONE on this one GPU with its own rocprofv3 --pmc passes (bench.py --tile R/N runs them itself), for the
# driver's arguments and the default ones.  profiles/update_traffic.py turns the lines into traffic.json entries keyed n_gpus = N.
# usage: bash profiles/pmc_rank_shapes.sh <tag>      (about 15 GPU-minutes)
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}_rank_shapes
mkdir -p $OUT
cd $ROOT
for ARGS in "--steps 20 --warmup 5" "--steps 16 --warmup 2"; do
  A=$(echo $ARGS | tr -d ' -')
  for T in 0/2 1/4 2/4 0/8 3/8 7/8; do
    python bench.py $ARGS --tile $T --no-cpu-baseline --no-also --no-forest > $OUT/tile_$(echo $T | tr / of)_$A.json 2> $OUT/tile_$(echo $T | tr / of)_$A.err
    echo "tile $T $ARGS: $(cut -c1-100 $OUT/tile_$(echo $T | tr / of)_$A.json)"
  done
done
