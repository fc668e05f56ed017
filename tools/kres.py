#!/usr/bin/env python3
"""Kernel resource usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel."""
import re, subprocess, sys
f = sys.argv[1]
extra = sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/tmp/kres.o"] + extra,
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{n:70s} vgpr {r.get('vgpr', -1):4d} agpr {r.get('agpr', -1):4d} scratch {r.get('scratch', -1):4d} occ {r.get('occ', -1)} lds {r.get('lds', -1)}")
