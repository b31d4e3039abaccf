#!/usr/bin/env python3
"""Rebuilds profiles/traffic.json from bench lines that carry live counters (roofline.pmc_counters):
one entry per launch shape (workload, image, frames per launch, kernel variant, GPUs).  bench.py
falls back to the matching entry when it cannot run its own counter passes (under a profiler).
usage: python profiles/update_traffic.py profiles/r03_g_*_bench*.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
entries = []
for path in sys.argv[1:]:
    with open(path) as f:
        j = json.loads([l for l in f if l.startswith("{")][-1])
    blocks = [(j["roofline"], j["config"]["workload"], j["config"]["image"])]
    if "forest" in j:       # the forest leg of the default run carries its own counter passes
        blocks.append((j["forest"]["roofline"], j["forest"]["workload"], [1920, 1080]))
    for r, name, image in blocks:
        if "pmc_counters" not in r or not str(r.get("traffic_source", "")).startswith("live"):
            continue
        workload = "demo" if "demo mesh" in name else ("forest" if "forest" in name else ("closeup" if "close-up" in name else "dragon"))
        key = {"workload": workload, "image": image, "frames_per_launch": r["frames_per_launch"], "variant": 0, "n_gpus": j["n_gpus"]}
        tile = j["config"].get("tile")
        if tile:
            # `bench.py --tile R/N`: rank R's share of an N-way split rendered alone on one GPU -- the shape an N-GPU line falls
            # back to (a rank cannot run counter passes inside the job).  One entry per (N, launch depth): the mean over the ranks measured.
            rk, n = (int(v) for v in tile.split("/"))
            key["n_gpus"] = n
            old = [e for e in entries if all(e[k] == v for k, v in key.items())]
            ranks = (old[0]["ranks_measured"] if old else []) + [rk]
            cnt = r["pmc_counters"]
            if old:
                k0 = len(old[0]["ranks_measured"])
                cnt = {c: (old[0]["counters"][c] * k0 + v) / (k0 + 1) for c, v in cnt.items() if c in old[0]["counters"]}
            entries = [e for e in entries if any(e[k] != v for k, v in key.items())]
            entries.append(dict(key, counters=cnt, kernel_ms=r["kernel_ms"], ranks_measured=ranks,
                                source=f"rank-of-{n} shape measured on one GPU (bench.py --tile R/{n}, ranks {ranks}): " + os.path.relpath(os.path.dirname(path), ROOT)))
            continue
        entries = [e for e in entries if any(e[k] != v for k, v in key.items())]
        entries.append(dict(key, counters=r["pmc_counters"], kernel_ms=r["kernel_ms"], source=os.path.relpath(path, ROOT)))
with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
    json.dump({"entries": entries}, f, indent=1)
print(f"{len(entries)} entries")
