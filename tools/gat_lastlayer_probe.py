#!/usr/bin/env python3
"""Output layer of the 8-head GAT on the Reddit-sized graph (64 hidden -> 8 heads x 41 classes, heads averaged):
aggregate-then-transform (ggl_gat_sh_*, 256 B gathered per edge) vs transform-then-aggregate (wide-head kernels,
1408 B per edge), with and without attention dropout; and the whole 2-layer GATModel step of config 3."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine, layers  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
x = torch.randn(n, 64, device=dev, requires_grad=True)


def ev(fn, reps=5):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for p in (0.0, 0.6):
    conv = layers.FusedGATConv(64, 41, heads=8, concat=False, dropout_rate=p).to(dev)
    conv.train()
    for fast in (True, False):
        eng.gat_fast = fast
        f = ev(lambda: conv(x.detach(), ei, n))
        fb = ev(lambda: conv(x, ei, n).sum().backward())
        print(f"FusedGATConv(64 -> 8 x 41, heads averaged) dropout {p} "
              f"{'aggregate-then-transform' if fast else 'transform-then-aggregate'}: fwd {f:.2f} ms "
              f"({E / f / 1e6:.1f} Gedges/s), fwd+bwd {fb:.2f} ms", flush=True)
eng.gat_fast = True
xf = torch.randn(n, 602, device=dev)
yl = torch.randint(0, 41, (n,), device=dev)
tidx = torch.arange(0, n, 3, device=dev)
for fast in (True, False):
    eng.gat_fast = fast
    torch.manual_seed(0)
    net = layers.GATModel(602, 8, 41, 8, 0.6, 2, fused=True).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=5e-4)

    def step():
        net.train()
        opt.zero_grad(set_to_none=True)
        F.cross_entropy(net(xf, ei, n)[tidx], yl[tidx]).backward()
        opt.step()

    ms = ev(step)
    print(f"2-layer 8-head GATModel(602 -> 8x8 -> 41) training step, dropout 0.6, "
          f"{'fast paths' if fast else 'round-1 kernels'}: {ms:.1f} ms ({4 * E / ms / 1e6:.2f} Gedges/s)", flush=True)
eng.gat_fast = True
