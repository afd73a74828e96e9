#!/usr/bin/env python3
"""Halo volumes of the 1-D node partition for P = 2, 4, 8 (single GPU, no comm): rows each rank must receive per
layer, bytes at K = 256, local-source share of the edges — for random ids, the generator's natural order and the
locality-aware order of gammagl_amd.partition.cluster_order, on (a) the R-MAT bench graph (no community
structure: nothing to find) and (b) a planted-community graph of the same size class (structure to find).

    python tools/halo_stats.py [products|arxiv]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.partition import cluster_order, halo_stats, relabel_edges  # noqa: E402
from gammagl_amd.synth import DATASETS, homophilous_graph, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "products"
n, e, _, _ = DATASETS[name]


def report(label, ei, n):
    for P in (2, 4, 8):
        h, share = halo_stats(ei, n, P)
        print(f"  {label:34s} P={P}: halo rows/rank max {h:9d} ({h * 1024 / 1e9:6.3f} GB at K=256), "
              f"local-source share {share:.2f}", flush=True)


def clustered(ei, n, clusters):
    eng.graph_cache.clear(); eng.seg_cache.clear()
    t0 = time.perf_counter()
    rank, lab = cluster_order(ei, n, clusters=clusters, sweeps=20, eng=eng)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sizes = torch.bincount(lab, minlength=clusters)
    print(f"  cluster_order({clusters} labels, 20 sweeps): {dt:.2f} s, community sizes {int(sizes.min())}..{int(sizes.max())}")
    return relabel_edges(ei, rank)


print(f"(a) {name}-sized R-MAT, N={n}")
ei = rmat_graph(n, e, seed=0, device=dev, relabel="random")
report("random ids", ei, n)
report("cluster_order on random ids", clustered(ei, n, 64), n)
del ei
ei = rmat_graph(n, e, seed=0, device=dev, relabel="none")
report("generator's natural order", ei, n)
del ei
deg = max(2, e // (2 * n))
print(f"(b) planted communities: N={n}, 64 classes, {deg} out-edges per node, 85 % inside the class, random ids")
_, y, ei = homophilous_graph(n, 64, 64, deg=deg, p_same=0.85, seed=0, device=dev)
pi = torch.randperm(n, device=dev)
ei = torch.stack([pi[ei[0]], pi[ei[1]]])
report("random ids", ei, n)
report("cluster_order on random ids", clustered(ei, n, 256), n)
inv = torch.empty_like(pi)
inv[pi] = torch.arange(n, device=dev)
rank_true = torch.empty(n, dtype=torch.int64, device=dev)
rank_true[torch.argsort(y[inv] * n + torch.arange(n, device=dev))] = torch.arange(n, device=dev)
report("oracle order (the planted classes)", relabel_edges(ei, rank_true), n)
