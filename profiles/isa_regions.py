#!/usr/bin/env python3
"""Static instruction counts of one k_raytrace_sm instantiation by source region (node step, triangle step, service parts).
Needs an assembly listing with line tables:
  hipcc --offload-arch=gfx950 -O3 ... -gline-tables-only -S --cuda-device-only -o g.s csrc/pt_kernels.hip
usage: python profiles/isa_regions.py g.s <mangled-name-substring> [line:label ...]
Instructions carry the location of the innermost inlined callee; one that lies outside the kernel body (a helper, pt_devmath.h)
is attributed to the region of the last kernel-body location seen before it -- approximate across scheduling, good to a few %."""
import collections, re, sys
path, needle = sys.argv[1], sys.argv[2]
marks = sorted((int(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[3:])
body_lo, body_hi = marks[1][0], marks[-1][0]      # (lines of the first region -- the prologue, where the stack lambdas are defined -- never switch)


def region(line):
    r = None
    for lo, name in marks:
        if line >= lo:
            r = name
    return r


lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2pt13k_raytrace_sm") and needle in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = marks[0][1]
counts = collections.Counter(); valu = collections.Counter(); vmem = collections.Counter(); lds = collections.Counter(); salu = collections.Counter()
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        f, ln = int(m.group(1)), int(m.group(2))
        if f == 0 and body_lo <= ln < body_hi:
            cur = region(ln)
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    counts[cur] += 1
    if op.startswith("v_"): valu[cur] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): vmem[cur] += 1
    elif op.startswith("ds_"): lds[cur] += 1
    elif op.startswith("s_"): salu[cur] += 1
print(f"{'region':28s} {'all':>6s} {'VALU':>6s} {'SALU':>6s} {'VMEM':>5s} {'LDS':>5s}")
for _, name in marks[:-1]:
    if name in counts:
        print(f"{name:28s} {counts[name]:6d} {valu[name]:6d} {salu[name]:6d} {vmem[name]:5d} {lds[name]:5d}")
print(f"{'total':28s} {sum(counts.values()):6d} {sum(valu.values()):6d} {sum(salu.values()):6d} {sum(vmem.values()):5d} {sum(lds.values()):5d}")
