#!/usr/bin/env python3
"""segment_max / gspmm max on the products-sized graph after the argmax witnesses moved to 32-bit registers, with the
ragged lanes for max off / on (A/B), sum beside it.   python tools/max_probe.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
dst = ei[1].contiguous()
w = torch.rand(E, device=dev)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def ev(fn, reps=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for dt in (torch.float32, torch.float16, torch.bfloat16):
    for K in (16, 47, 64, 100):
        x = (torch.randn(E, K, device=dev) * 4).to(dt)
        ts = ev(lambda: eng.c_segment_sum(x, dst, n))
        res = []
        for rm in (0, 1):
            eng.set_option("ragged_max", rm)
            res.append(ev(lambda: eng.c_segment_max(x, dst, n)))
        eng.set_option("ragged_max", 0)
        say(f"segment {str(dt)[6:]:9s} K={K:3d}: sum {ts:6.2f} ms | max {res[0]:6.2f} ms, with ragged lanes {res[1]:6.2f} ms")
        del x
for K in (47, 100, 256):
    x = torch.randn(n, K, device=dev)
    say(f"gspmm f32 K={K:3d}: sum {ev(lambda: eng.c_spmm_sum(ei, w, x)):6.2f} ms | max {ev(lambda: eng.c_spmm_max(ei, w, x)):6.2f} ms")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
