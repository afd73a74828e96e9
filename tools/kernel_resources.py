#!/usr/bin/env python3
"""Print VGPR / SGPR / occupancy / scratch of every kernel in a .hip file (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"],
                   capture_output=True, text=True)
cur, d = None, {}
for ln in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        d = {}
    for k in ("VGPRs", "AGPRs", "SGPRs", "Occupancy", "ScratchSize"):
        m = re.search(r" " + k + r"[^:]*: (\d+)", ln)
        if m:
            d[k] = int(m.group(1))
    if "LDS Size" in ln and cur and flt in cur:
        print(f"{cur[:110]:110s} {d}")
