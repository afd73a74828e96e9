#!/usr/bin/env python3
"""How much reuse the bench graph offers a cache of k source rows: share of the edges whose SOURCE is among the k
highest-degree nodes (products-sized R-MAT, CPU, ~1 min).  A 64-column launch gathers 256-byte slices: one XCD's 4 MiB L2
holds 16 384 of them, the 256 MiB Infinity Cache 1 M.   python tools/hub_share_probe.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.synth import rmat_graph, DATASETS
torch.set_num_threads(8)
n, e, _, _ = DATASETS["products"]
t0 = time.time()
ei = rmat_graph(n, e, seed=0, device="cpu")
print("built", ei.shape, time.time() - t0)
deg = torch.bincount(ei[0], minlength=n)
d, _ = torch.sort(deg, descending=True)
c = torch.cumsum(d, 0).double() / d.sum()
for k in (1024, 4096, 16384, 65536, 262144, 1048576):
    print(f"top {k:8d} sources cover {float(c[k-1]):.3f} of the edges; degree at rank k: {int(d[k-1])}")
