#!/usr/bin/env python3
"""Fused GAT on the Reddit-sized graph with and without attention dropout (forward, forward+backward)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
x = torch.randn(n, 8, 8, device=dev, requires_grad=True)
el = torch.randn(n, 8, device=dev, requires_grad=True)
er = torch.randn(n, 8, device=dev, requires_grad=True)


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


for p in (0.0, 0.6):
    f = ev(lambda: eng.gat_fused(ei, el.detach(), er.detach(), x.detach(), 0.2, dropout_rate=p))
    fb = ev(lambda: eng.gat_fused(ei, el, er, x, 0.2, dropout_rate=p).sum().backward())
    print(f"gat_fused reddit-sized (E={ei.shape[1]}, H=8, C=8) dropout_rate={p}: fwd {f:.2f} ms, fwd+bwd {fb:.2f} ms")
