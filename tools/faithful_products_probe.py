#!/usr/bin/env python3
"""The products-sized 3-layer GCN step through the layer classes as the reference writes them
(GCNModel / GCNConv(norm='both'): degrees and edge weights recomputed in every layer, ReLU and dropout as
separate ops) vs the harness bench.py times (weights precomputed, fused epilogue, side-stream weight
gradients).  Same kernels underneath; the difference is what a user gains by adopting the fused entry points."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402
from gammagl_amd.trainer import GCNTrainer  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, f, c = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, f, generator=g, device=dev)
y = torch.randint(0, c, (n,), generator=g, device=dev)
idx = torch.nonzero(torch.rand(n, generator=g, device=dev) < 0.08).reshape(-1)
for norm in ("both", "none"):
    tr = GCNTrainer(f, 256, c, num_layers=3, norm=norm, device=dev)
    for _ in range(3):
        tr.step(x, ei, y, idx, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        tr.step(x, ei, y, idx, n)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    print(f"GCNModel / GCNConv(norm='{norm}') as written in the reference: {ms:.1f} ms/step ({6 * E / ms / 1e6:.2f} Gedges/s)")
