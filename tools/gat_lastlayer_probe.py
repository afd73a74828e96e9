#!/usr/bin/env python3
"""Last layer of the 8-head GAT on the Reddit-sized graph: 8 heads x 41 classes (K = 328, C % 4 != 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import layers  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
x = torch.randn(n, 64, device=dev, requires_grad=True)


def ev(fn, reps=3):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for C in (41, 44, 40):
    conv = layers.FusedGATConv(64, C, heads=8, concat=False).to(dev)
    f = ev(lambda: conv(x.detach(), ei, n))
    fb = ev(lambda: conv(x, ei, n).sum().backward())
    print(f"FusedGATConv(64 -> 8 heads x {C}, mean over heads) on E={ei.shape[1]}: fwd {f:.2f} ms, fwd+bwd {fb:.2f} ms", flush=True)
