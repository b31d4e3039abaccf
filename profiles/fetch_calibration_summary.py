#!/usr/bin/env python3
"""Table for profiles/fetch_calibration.sh: per pattern, the known byte count, each counter's value and bytes / (FETCH_SIZE KB).
The probe's dispatches are identified by their order (it prints one line per pattern in that order)."""
import collections, csv, glob, re, sys
root = sys.argv[1]
names, known, ms = [], {}, {}
for line in open(f"{root}/timing.txt"):
    m = re.match(r"(\S+)\s+requested\s+(\d+) bytes\s+([\d.]+) ms", line)
    if m:
        names.append(m.group(1)); known[m.group(1)] = float(m.group(2)); ms[m.group(1)] = float(m.group(3))
# dispatches per pattern: one each, except the small-table streaming sweeps (32 launches)
per_pattern = [32 if n.startswith("stream16_small") else 1 for n in names]
vals = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{root}/pass*/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_gather" in r["Kernel_Name"] or "k_stream16" in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    for c, d in per.items():
        ordered = [d[k] for k in sorted(d)]
        at = 0
        for n, k in zip(names, per_pattern):
            vals[n][c] = sum(ordered[at:at + k])
            at += k
cols = sorted({c for v in vals.values() for c in v})
print(f"{'pattern':22s} {'known bytes':>14s} {'ms':>8s} {'GB/s':>8s} " + " ".join(f"{c:>22s}" for c in cols) + "   bytes/(FETCH_SIZE*1024)  bytes/(RDREQ*64)  bytes/(32*n32+64*n64+128*n128)")
for n in names:
    v = vals.get(n, {})
    fs = v.get("FETCH_SIZE")
    rq = v.get("TCC_EA0_RDREQ_sum")
    print(f"{n:22s} {known[n]:14.0f} {ms[n]:8.3f} {known[n] / ms[n] / 1e6:8.1f} " + " ".join(f"{v.get(c, float('nan')):22.0f}" for c in cols)
          + (f"   {known[n] / (fs * 1024):10.3f}" if fs else "          n/a") + (f"   {known[n] / (rq * 64):10.3f}" if rq else "          n/a")
          + (f"   {known[n] / (32 * v.get('TCC_EA0_RDREQ_32B_sum', 0) + 64 * v['TCC_EA0_RDREQ_64B_sum'] + 128 * v['TCC_EA0_RDREQ_128B_sum']):10.3f}"
             if v.get("TCC_EA0_RDREQ_64B_sum") is not None and v.get("TCC_EA0_RDREQ_128B_sum") is not None and (v["TCC_EA0_RDREQ_64B_sum"] + v["TCC_EA0_RDREQ_128B_sum"]) > 0 else "          n/a"))
