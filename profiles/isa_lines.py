#!/usr/bin/env python3
"""Static instruction counts per SOURCE LINE of one kernel of a -gline-tables-only listing (see isa_regions.py).
usage: python profiles/isa_lines.py g.s <mangled-name-substring> <first line> <last line> [more ranges first last ...]"""
import re, collections, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path, needle = sys.argv[1], sys.argv[2]
ranges = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(3, len(sys.argv) - 1, 2)]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2pt") and needle in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = (0, 0); c = collections.Counter(); v = collections.Counter()
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); continue
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    c[cur] += 1
    if s.startswith("v_"): v[cur] += 1
src = open(os.path.join(ROOT, "webgpu-pathtracer_amd", "csrc", "pt_kernels.hip")).read().split("\n")
tot = totv = 0
for (f, ln), n in sorted(c.items()):
    if f == 0 and any(a <= ln <= b for a, b in ranges):
        tot += n; totv += v[(f, ln)]
        print(f"{ln:5d} all {n:3d} valu {v[(f, ln)]:3d}  {src[ln - 1].strip()[:120]}")
print(f"total all {tot} valu {totv}")
