#!/usr/bin/env python3
"""K = 64 / 128 SpMM-sum on the products-sized graph: float4 lane groups (4 or 2 rows per wave, vector index
loads) vs one wave per row with dword lanes (scalar index loads) — A/B through the force_generic option."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.layers import calc_gcn_norm  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS[sys.argv[1] if len(sys.argv) > 1 else "products"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
w = calc_gcn_norm(ei, n).contiguous()
gp = eng.graph_plan(ei, n)
for K in [int(k) for k in os.environ.get("KS", "32,64,128").split(",")]:
    x = torch.randn(n, K, device=dev)
    line = f"K={K:4d}:"
    for fg in (0, 1):
        eng.set_option("force_generic", fg)
        ms = eng.time_spmm_sum(gp, w, x, reps=10)
        line += f"  force_generic={fg}: {ms:.3f} ms ({(E * (4 * K + 8) + n * (4 * K + 8)) / ms / 1e9:.2f} TB/s alg)"
    print(line, flush=True)
eng.set_option("force_generic", 0)
