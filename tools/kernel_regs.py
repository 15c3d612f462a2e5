#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in one .hip source (hipcc -Rpass-analysis=kernel-resource-usage,
gfx950 cross-compile, no GPU needed).   python tools/kernel_regs.py csrc/elementwise.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        # light demangling of the template arguments (c++filt here predates the _Float16 mangling)
        t = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?_kernel)(I.*?E)?Ev", name)
        if t:
            args = re.findall(r"L[ib](\d+)E", t.group(2) or "")
            name = t.group(1) + ("<" + ", ".join(args) + ">" if args else "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':90s} vgpr agpr spill scratch occ lds")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:90]:90s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('spill', 0):5d} {r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('lds', 0)}")
