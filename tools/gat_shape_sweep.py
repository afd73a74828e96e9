#!/usr/bin/env python3
"""Fused GAT op over head-count x channel shapes on the Reddit-sized graph: ms forward, forward+backward, and
the forward against its HBM roofline (E * (4HC + 4H + 4) algorithmic bytes) — looking for shape cliffs."""
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
E = ei.shape[1]


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


shapes = [(1, 8), (1, 64), (1, 256), (2, 128), (4, 64), (8, 32), (8, 16), (16, 8), (16, 16), (3, 24), (8, 41), (6, 7)]
for H, C in shapes:
    x = torch.randn(n, H, C, device=dev, requires_grad=True)
    el = torch.randn(n, H, device=dev, requires_grad=True)
    er = torch.randn(n, H, device=dev, requires_grad=True)
    f = ev(lambda: eng.gat_fused(ei, el.detach(), er.detach(), x.detach(), 0.2))
    fb = ev(lambda: eng.gat_fused(ei, el, er, x, 0.2).sum().backward())
    alg = E * (4 * H * C + 4 * H + 4)
    print(f"H={H:2d} C={C:3d} (K={H * C:3d}): fwd {f:7.2f} ms = {alg / f / 1e9:5.2f} TB/s alg   fwd+bwd {fb:8.2f} ms "
          f"(bwd/fwd {((fb - f) / f):4.1f}x)", flush=True)
    del x, el, er
