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
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "products"
n, e, _, _ = DATASETS[name]


def report(label, ei, n):
    for P in (2, 4, 8):
        h, share = halo_stats(ei, n, P)
        print(f"  {label:34s} P={P}: halo rows/rank max {h:9d} ({h * 1024 / 1e9:6.3f} GB at K=256), "
              f"local-source share {share:.2f}", flush=True)


def clustered(ei, n, clusters, sweeps=20):
    eng.graph_cache.clear(); eng.seg_cache.clear()
    t0 = time.perf_counter()
    rank, lab = cluster_order(ei, n, clusters=clusters, sweeps=sweeps, eng=eng)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sizes = torch.bincount(lab, minlength=clusters)
    print(f"  cluster_order({clusters} labels, {sweeps} sweeps): {dt:.2f} s, community sizes {int(sizes.min())}..{int(sizes.max())}")
    return relabel_edges(ei, rank)


print(f"(a) {name}-sized R-MAT, N={n}")
ei = rmat_graph(n, e, seed=0, device=dev, relabel="random")
report("random ids", ei, n)
report("cluster_order on random ids", clustered(ei, n, 64), n)
del ei
ei = rmat_graph(n, e, seed=0, device=dev, relabel="none")
report("generator's natural order", ei, n)
del ei
from gammagl_amd.synth import PLANTED_LEVELS, planted_pairs  # noqa: E402

deg = max(2, e // (2 * n))
print(f"(b) hierarchical planted communities: N={n}, levels (groups, share of a node's edges inside) = {PLANTED_LEVELS}, "
      f"{deg} out-edges per node, random ids")
s_, d_ = planted_pairs(n, out_deg=deg, seed=0, device=dev)
nat = torch.stack([s_, d_])
pi = torch.randperm(n, device=dev)
ei = torch.stack([pi[s_], pi[d_]]).contiguous()
del s_, d_
report("random ids", ei, n)
report("cluster_order on random ids", clustered(ei, n, max(8, min(8192, n // 600)), sweeps=30), n)
report("oracle order (the generator's own labelling)", nat, n)
