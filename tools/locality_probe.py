#!/usr/bin/env python3
"""What a locality-aware node order buys the aggregate itself (one GPU): SpMM-sum at K = 16 / 64 / 256 on a
products-sized planted-community graph with random ids vs the ids of partition.cluster_order — narrow rows are bound
by the 128-byte line, so rows of neighbours sharing lines / L2 is the only lever left."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.partition import cluster_order, relabel_edges  # noqa: E402
from gammagl_amd.synth import DATASETS, homophilous_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["products"]
deg = max(2, e // (2 * n))
_, y, ei = homophilous_graph(n, 64, 64, deg=deg, p_same=0.85, seed=0, device=dev)
pi = torch.randperm(n, device=dev)
ei = torch.stack([pi[ei[0]], pi[ei[1]]]).contiguous()
E = ei.shape[1]
print(f"planted-community graph N={n} E={E} (64 classes, 85 % of the edges inside the class, ids shuffled)")


def timed(fn, reps=9):
    for _ in range(3):
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


rank, lab = cluster_order(ei, n, clusters=256, sweeps=20, eng=eng)
ei_c = relabel_edges(ei, rank).contiguous()
eng.clear_caches()
for label, edges in (("random ids    ", ei), ("cluster_order ", ei_c)):
    gp = eng.graph_plan(edges, n)
    for K in (16, 64, 256):
        x = torch.randn(n, K, device=dev)
        with torch.no_grad():
            t = timed(lambda: eng.spmm(gp, None, x))
        alg = E * (4 * K + 4) + n * (4 * K + 8)
        print(f"{label} K={K:3d}: {t:7.3f} ms  {E / t / 1e6:6.1f} Gedges/s  {alg / t / 1e6:6.0f} GB/s algorithmic", flush=True)
    eng.clear_caches()
