#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel: tools/isa_blocks.py file.s 'demangled substring' [min_vmem]."""
import collections
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
min_vmem = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for m in re.finditer(r"^(_Z\w+):.*$", s, re.M):
    name = m.group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0]
    if want not in dem:
        continue
    i = m.end()
    j = s.index(".Lfunc_end", i)
    blocks = []
    blk = ["entry", collections.Counter()]
    blocks.append(blk)
    for ln in s[i:j].splitlines():
        t = ln.strip()
        if t.startswith(".LBB") and t.endswith(":"):
            blk = [t, collections.Counter()]
            blocks.append(blk)
        elif t and not t.startswith(";") and not t.startswith("."):
            op = t.split()[0]
            cls = ("valu" if op.startswith("v_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else
                   "smem" if op.startswith("s_load") else "wait" if op.startswith("s_waitcnt") else
                   "lds" if op.startswith("ds_") else "salu")
            blk[1][cls] += 1
            if op.startswith("v_exp") or "dpp" in t:
                blk[1]["exp" if op.startswith("v_exp") else "dpp"] += 1
    k = s.find(".name:           " + name)
    vg = re.search(r"\.vgpr_count:\s+(\d+)", s[k:k + 4000]) if k >= 0 else None
    print(dem, "vgpr", vg.group(1) if vg else "?")
    for b in blocks:
        if b[1]["vmem"] >= min_vmem:
            print("   ", b[0], dict(b[1]))
