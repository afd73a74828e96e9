#!/usr/bin/env python3
"""tools/pmc_probe.py — a few launches of the dominant kernel (CSR SpMM-sum, K=256) on the bench
graph, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE are collected in separate passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_partitioned  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "products"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
relabel = sys.argv[4] if len(sys.argv) > 4 else "random"
order = sys.argv[5] if len(sys.argv) > 5 else "src"
dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS[wl]
g = rmat_partitioned(n, e, seed=seed, device=dev, relabel=relabel, order=order)   # the bench's own graph
ei = torch.stack([g["src"], g["dst"]])
w = g["w"]
gp = eng.graph_plan(ei, n)
x = torch.randn(n, K, device=dev)
torch.cuda.synchronize()
eng._sorted_weights(gp.fwd, w)
for _ in range(4):
    eng._spmm_fwd("sum", gp.fwd, gp.col, w, x, n)
torch.cuda.synchronize()
print("E", ei.shape[1], "N", n, "K", K, "alg_bytes", ei.shape[1] * (4 * K + 8) + n * (4 * K + 8))
