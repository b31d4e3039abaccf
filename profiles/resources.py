#!/usr/bin/env python3
"""Register / spill / scratch / LDS / occupancy of every kernel in pt_kernels.hip, one line each
(`make -C webgpu-pathtracer_amd/csrc resources` piped through c++filt).
usage: python profiles/resources.py [extra hipcc flags, e.g. -DMI3PT_EXPERIMENTS]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "webgpu-pathtracer_amd", "csrc")
flags = ("-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt "
         "-Wno-unused-function").split()
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags, *sys.argv[1:], "-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(csrc, "pt_kernels.hip"), "-o", "/dev/null"]
text = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = ("VGPRs", "SGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k in keys:
        m = re.search(r"remark: [^:]*:\d+:\d+: +" + re.escape(k) + r": (\d+)", line) or re.search(r"\s" + re.escape(k) + r": (\d+)\s*\[", line)
        if m and cur is not None and k not in cur:
            cur[k] = m.group(1)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
n_sm = 0
for r, n in zip(rows, names):
    n = n.replace("void pt::", "").replace("(pt::RtLaunch)", "")
    n_sm += "k_raytrace_sm" in n
    print(f"{n[:96]:96s} VGPR {r.get('VGPRs', '?'):>3s}  SGPR {r.get('SGPRs', '?'):>3s}  sgpr-spill {r.get('SGPRs Spill', '?'):>3s}  vgpr-spill {r.get('VGPRs Spill', '?'):>3s}"
          f"  scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s}  occ {r.get('Occupancy [waves/SIMD]', '?'):>2s}  lds {r.get('LDS Size [bytes/block]', '?'):>5s}")
print(f"{len(rows)} kernels, {n_sm} k_raytrace_sm instantiations")
