#!/usr/bin/env python3
"""Format-conversion entry points (ops/sparse ind2ptr / ptr2ind, utils.sort_edge_index) and the plan build at the
products size: ms per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine, sparse  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS[sys.argv[1] if len(sys.argv) > 1 else "products"]
ei = rmat_graph(n, e, seed=0, device=dev, order="src")
E = ei.shape[1]


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"N={n} E={E}")
print(f"sort_edge_index (by row):     {t(lambda: sparse.sort_edge_index(ei)):8.2f} ms")
print(f"sort_edge_index (by col):     {t(lambda: sparse.sort_edge_index(ei, sort_by_row=False)):8.2f} ms")
srt = sparse.sort_edge_index(ei)
srt = srt[0] if isinstance(srt, tuple) else srt
ptr = sparse.ind2ptr(srt[0], n)
print(f"ind2ptr:                      {t(lambda: sparse.ind2ptr(srt[0], n)):8.2f} ms")
print(f"ptr2ind:                      {t(lambda: sparse.ptr2ind(ptr, E)):8.2f} ms")


def plan():
    eng.seg_cache.clear(); eng.graph_cache.clear()
    gp = eng.graph_plan(ei.clone(), n)
    gp.bwd, gp.colT, gp.posT  # noqa: B018


print(f"graph plan (fwd + transposed + posT) from an unsorted COO: {t(plan):8.2f} ms")
