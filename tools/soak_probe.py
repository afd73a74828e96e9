#!/usr/bin/env python3
"""Soak: 60 products-sized training steps; memory must not grow, step time must not drift, loss must fall."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.dist import DistGCNTrainer, PartitionedGraph  # noqa: E402
from gammagl_amd.layers import calc_gcn_norm  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, f, c = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
w = calc_gcn_norm(ei, n).contiguous()
pg = PartitionedGraph(ei, w, n, 0, 1, eng=eng)
E = ei.shape[1]
del ei, w
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, f, generator=g, device=dev)
y = torch.randint(0, c, (n,), generator=g, device=dev)
idx = torch.nonzero(torch.rand(n, generator=g, device=dev) < 0.08).reshape(-1)
tr = DistGCNTrainer(pg, f, 256, c, num_layers=3, seed=0, device=dev)
for block in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss = tr.step(x, y, idx, idx.numel())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"steps {block * 10 + 1:2d}-{block * 10 + 10:2d}: {ms:6.2f} ms/step  loss {float(loss):.4f}  allocated "
          f"{torch.cuda.memory_allocated() / 1e9:.2f} GB  reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB  "
          f"plans built {eng.stats['plans_built']}", flush=True)
