#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs produced by pmc_passes.sh: per-launch mean of every
counter for kernels whose name contains the given substring."""
import collections, csv, glob, sys
root, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_raytrace")
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f"{root}/pass*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:40s} {sum(v)/len(v):18.1f}   (n={len(v)})")
