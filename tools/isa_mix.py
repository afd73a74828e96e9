#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only): per kernel, instruction counts by
class inside its LONGEST loop-free... no: whole body and the hottest loop (the basic blocks between the first backward branch
target with the most instructions).  Usage: isa_mix.py file.s [name filter]"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""


def cls(op):
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith("v_mfma"): return "mfma"
    if "dpp" in op: return "dpp"
    if op.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac")): return "v_fp"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith(("v_exp", "v_rcp", "v_log")): return "trans"
    if op.startswith("v_"): return "v_other"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")): return "vmem_st"
    if op.startswith("ds_"): return "lds"
    return "other"


for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", src, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    lines = body.splitlines()
    # basic blocks: label -> index
    labels = {}
    ins = []
    for ln in lines:
        lm = re.match(r"^(\.LBB\w+):", ln)
        if lm:
            labels[lm.group(1)] = len(ins)
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")):
            continue
        ins.append(t)
    # loops: a branch at index i to a label at index j <= i
    best = None
    for i, t in enumerate(ins):
        bm = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", t)
        if bm and bm.group(1) in labels and labels[bm.group(1)] <= i:
            j = labels[bm.group(1)]
            if best is None or i - j > best[1] - best[0]:
                best = (j, i)
    tot = Counter(cls(t.split()[0]) for t in ins)
    print(name[:100])
    print("   whole:", len(ins), dict(tot))
    if best:
        lp = Counter(cls(t.split()[0]) for t in ins[best[0]:best[1] + 1])
        valu = sum(v for k, v in lp.items() if k in ("v_pk", "dpp", "v_fp", "cndmask", "trans", "v_other"))
        print(f"   longest loop: {best[1] - best[0] + 1} instrs, VALU {valu}:", dict(lp))
