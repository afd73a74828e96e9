#!/usr/bin/env python3
"""Halo volumes of the 1-D node partition on the bench graph for P = 2, 4, 8 (single GPU, no comm):
rows each rank must receive per layer, bytes at K = 256, local/halo edge split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.dist import balanced_bounds  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
n, e, _, _ = DATASETS[sys.argv[1] if len(sys.argv) > 1 else "products"]
for relabel in ("random", "none"):
    ei = rmat_graph(n, e, seed=0, device=dev, relabel=relabel)
    src, dst = ei[0], ei[1]
    for P in (2, 4, 8):
        b = balanced_bounds(dst, n, P)
        halo, loc_e, tot_e = [], [], []
        for r in range(P):
            lo, hi = b[r], b[r + 1]
            m = (dst >= lo) & (dst < hi)
            s = src[m]
            rem = s[(s < lo) | (s >= hi)]
            halo.append(int(torch.unique(rem).numel()))
            loc_e.append(int(m.sum()) - int(rem.numel()))
            tot_e.append(int(m.sum()))
        print(f"relabel={relabel:6s} P={P}: rows/rank {[b[i + 1] - b[i] for i in range(P)][:3]}.. halo rows/rank max {max(halo)} "
              f"({max(halo) * 1024 / 1e9:.2f} GB at K=256), edges/rank max {max(tot_e)}, local-source share "
              f"{sum(loc_e) / sum(tot_e):.2f}", flush=True)
    del ei
