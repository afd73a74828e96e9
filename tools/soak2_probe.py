#!/usr/bin/env python3
"""Soak of the round-2 paths: 500 replays of the graph-captured GraphSAGE mini-batch step (memory flat, no capacity
overflow, loss falls on a learnable graph) and 30 steps of the 2-layer GAT model on the Reddit-sized graph (memory
flat, step time steady)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine, layers  # noqa: E402
from gammagl_amd.sampler import BlockSampler  # noqa: E402
from gammagl_amd.synth import DATASETS, homophilous_graph, rmat_graph  # noqa: E402
from gammagl_amd.trainer import SAGEBlockTrainer  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n = 1_000_000
x, y, ei = homophilous_graph(n, 64, 16, deg=12, p_same=0.7, signal=0.4, seed=0, device=dev)
bs = BlockSampler(ei, [25, 10], num_nodes=n, eng=eng)
B = 2048
caps = bs.calibrate(B, trials=8, slack=1.3)
tr = SAGEBlockTrainer(bs, 64, 256, 16, lr=0.003, device=dev, caps=caps)
g = torch.Generator(device=dev).manual_seed(0)
seeds = torch.randperm(n, generator=g, device=dev)[:B].contiguous()
tr.capture(x, y, seeds)
for block in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ls = []
    for _ in range(100):
        seeds.copy_(torch.randint(0, n, (B,), generator=g, device=dev))
        ls.append(tr.replay().clone())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 100 * 1e3
    print(f"SAGE hipGraph replays {block * 100 + 1:3d}-{block * 100 + 100:3d}: {ms:5.2f} ms/batch  loss {float(torch.stack(ls).mean()):.4f}  "
          f"allocated {torch.cuda.memory_allocated() / 1e9:.2f} GB  reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB  "
          f"overflowed hops {bs.overflow_count()}", flush=True)
del tr, bs, x, y, ei
eng.clear_caches()
torch.cuda.empty_cache()
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
xf = torch.randn(n, 602, device=dev)
yl = torch.randint(0, 41, (n,), device=dev)
tidx = torch.arange(0, n, 3, device=dev)
net = layers.GATModel(602, 8, 41, 8, 0.6, 2, fused=True).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=5e-4)
for block in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        net.train()
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(net(xf, ei, n)[tidx], yl[tidx])
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"GAT model steps {block * 10 + 1:2d}-{block * 10 + 10:2d}: {ms:6.2f} ms/step  loss {float(loss):.4f}  allocated "
          f"{torch.cuda.memory_allocated() / 1e9:.2f} GB  reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB  plans built "
          f"{eng.stats['plans_built']}", flush=True)
