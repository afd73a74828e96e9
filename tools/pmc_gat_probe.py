#!/usr/bin/env python3
"""A few launches of the fused GAT forward + backward on the Reddit-sized graph, for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE collected in separate passes, see tools/pmc_summary.py)."""
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
H, C = 8, 8
x = torch.randn(n, H, C, device=dev, requires_grad=True)
el = torch.randn(n, H, device=dev, requires_grad=True)
er = torch.randn(n, H, device=dev, requires_grad=True)
for _ in range(3):
    eng.gat_fused(ei, el, er, x, 0.2).sum().backward()
torch.cuda.synchronize()
# the head-averaging output layer (64 -> 8 x 41, attention dropout 0.6): ggl_gat_sh_* (round 5: its source walk's counters)
from gammagl_amd import layers  # noqa: E402
conv = layers.FusedGATConv(64, 41, heads=8, concat=False, dropout_rate=0.6).to(dev)
conv.train()
xo = torch.randn(n, 64, device=dev, requires_grad=True)
for _ in range(3):
    conv(xo, ei, n).sum().backward()
torch.cuda.synchronize()
E = ei.shape[1]
print("E", E, "fwd alg bytes", E * (4 * H * C + 4 * H + 4) + n * (4 * H * C + 8 * H + 8))
