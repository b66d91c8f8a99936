#!/usr/bin/env python3
"""Instruction counts per source line of one loop of a kernel, from `hipcc -S -gline-tables-only` output:
   hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -gline-tables-only FILE.hip -o out.s
   python tools/isa_line_profile.py out.s FIRST_LINE LAST_LINE [SOURCE.hip]
FIRST/LAST: line numbers of the .s file that bracket the loop (see `grep -n "Loop Header" out.s`).  Prints VALU /
DPP / bpermute / LDS / VMEM / SALU counts per source line (innermost inlined location), largest first."""
import collections
import re
import sys

s = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])
src = open(sys.argv[4]).read().split("\n") if len(sys.argv) > 4 else None
files = {}
for line in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur = None
cnt = collections.defaultdict(lambda: collections.Counter())
tot = collections.Counter()
for i in range(lo - 1, hi):
    line = s[i].strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not line or line.startswith((".", ";")) or line.endswith(":"):
        continue
    op = line.split()[0]
    if op.startswith("v_"):
        kind = "dpp" if "dpp" in line else "valu"
    elif op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        kind = "bperm"
    elif op.startswith("ds_"):
        kind = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        kind = "vmem"
    elif op.startswith("s_"):
        kind = "salu"
    else:
        kind = "other"
    cnt[cur][kind] += 1
    tot[kind] += 1
print("total", dict(tot))
rows = sorted(cnt.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["dpp"] + kv[1]["bperm"]))
for (f, ln), c in rows[:int(sys.argv[5]) if len(sys.argv) > 5 else 60]:
    text = src[ln - 1].strip()[:90] if src and f and f.endswith(".hip") and ln <= len(src) and sys.argv[4].endswith(f) else ""
    print(f"{f}:{ln:5d}  valu {c['valu']:4d} dpp {c['dpp']:4d} bperm {c['bperm']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} salu {c['salu']:3d}  {text}")
