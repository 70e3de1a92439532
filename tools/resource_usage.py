#!/usr/bin/env python3
"""Condense `make -C utmos_amd/csrc asm`'s resource_usage.txt (hipcc -Rpass-analysis=kernel-resource-usage) into one line per
kernel:  python3 tools/resource_usage.py > profiles/rNN_resource_usage.txt"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "utmos_amd/csrc/resource_usage.txt")
rows, cur = [], None
keys = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds",
        "ScratchSize [bytes/lane]": "scratch", "SGPRs Spill": "ss", "VGPRs Spill": "vs"}
for ln in open(path):
    m = re.search(r"remark:\s+Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
    if m and cur is not None and m.group(1).strip() in keys:
        cur[keys[m.group(1).strip()]] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
tag = sys.argv[1] if len(sys.argv) > 1 else "final binary"
print(f"# hipcc -Rpass-analysis=kernel-resource-usage (make -C utmos_amd/csrc asm; tools/resource_usage.py), gfx950, {tag}; names demangled by c++filt")
print(f"{'kernel':80s} {'SGPR':>5s} {'VGPR':>5s} {'occ':>4s} {'LDS B':>7s} {'scratch':>8s} {'spillS':>6s} {'spillV':>6s}")
seen = set()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void ", ""))
    if n in seen or "sgpr" not in r:
        continue
    seen.add(n)
    print(f"{n[:80]:80s} {r.get('sgpr', 0):5d} {r.get('vgpr', 0):5d} {r.get('occ', 0):4d} {r.get('lds', 0):7d} {r.get('scratch', 0):8d} {r.get('ss', 0):6d} {r.get('vs', 0):6d}")
