#!/usr/bin/env python3
"""The reference's FusedGAT table (examples/fusedgat/readme.md:45-53: GAT vs FusedGAT, train and infer ms/epoch on
cora / citeseer / pubmed, hardware unstated: 20.4 -> 10.1 / 21.2 -> 10.0 / 20.4 -> 9.1 ms train, 4.0 -> 2.1 / 4.3 ->
2.2 / 4.2 -> 2.1 ms infer) re-measured here on synthetic graphs of those sizes: 2-layer GAT, hidden 8, 8 heads,
full-batch epoch = one training step."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.layers import GATModel, add_self_loops  # noqa: E402
from gammagl_amd.synth import homophilous_graph  # noqa: E402

dev = torch.device("cuda", 0)
SIZES = {"cora": (2708, 1433, 7, 2, 0.7), "citeseer": (3327, 3703, 6, 2, 0.6), "pubmed": (19717, 500, 3, 3, 0.2)}
for name, (n, f, c, deg, drop) in SIZES.items():
    x, y, ei = homophilous_graph(n, f, c, deg=deg, seed=0, device=dev)
    ei = add_self_loops(ei, n)
    idx = torch.arange(0, n, 10, device=dev)
    row = f"{name:9s} N={n:6d} E={ei.shape[1]:6d}:"
    for fused in (False, True):
        torch.manual_seed(0)
        net = GATModel(f, 8, c, 8, drop, 2, fused=fused).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=0.01, weight_decay=5e-3)

        def train():
            net.train()
            opt.zero_grad(set_to_none=True)
            F.cross_entropy(net(x, ei, n)[idx], y[idx]).backward()
            opt.step()

        def infer():
            net.eval()
            with torch.no_grad():
                net(x, ei, n)

        res = []
        for fn in (train, infer):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 50 * 1e3)
        row += f"  {'FusedGAT' if fused else 'GAT     '} train {res[0]:6.2f} ms  infer {res[1]:5.2f} ms  (peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MB);"
        torch.cuda.reset_peak_memory_stats()
    print(row, flush=True)
