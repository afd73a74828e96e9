#!/usr/bin/env python3
"""FusedGATConv forward+backward on the Reddit-sized graph, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine, layers  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
fg = layers.FusedGATConv(602, 8, heads=8).to(dev)
x = torch.randn(n, 602, device=dev)
for _ in range(4):
    xx = x.clone().requires_grad_(True)
    fg(xx, ei, n).sum().backward()
torch.cuda.synchronize()
print("done", ei.shape)
