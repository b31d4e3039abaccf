#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs produced by pmc_passes.sh: per-launch mean of every counter for
kernels whose name contains the given substring (argv[2], default k_raytrace), over the LAST
argv[3] dispatches of each pass when given (= the timed launches; the warm-up ones come first)."""
import collections, csv, glob, sys
root, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_raytrace")
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f"{root}/pass*/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, rows in per.items():
        rows.sort()
        agg[k] += [v for _, v in (rows[-last:] if last else rows)]
for k in sorted(agg):
    v = agg[k]
    print(f"{k:40s} {sum(v)/len(v):18.1f}   (n={len(v)})")
