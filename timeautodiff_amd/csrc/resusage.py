#!/usr/bin/env python3
"""Print a per-kernel table (VGPRs, spills, scratch, LDS, occupancy) for one .hip file.
usage: python resusage.py rnn.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_resusage.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows = []
cur = None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        cur = {"name": name}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                     ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':70s} vgpr agpr sgpr vspill scratch lds occ")
for r in rows:
    print(f"{r['name'][:70]:70s} {r.get('vgpr', -1):4d} {r.get('agpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('vspill', -1):6d} "
          f"{r.get('scratch', -1):7d} {r.get('lds', -1):5d} {r.get('occ', -1):3d}")
