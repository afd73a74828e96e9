#!/usr/bin/env python3
"""Full-graph SAGEConv(mean) layer (GraphSAGE_Full_Model, models/graphsage.py:7-32) on the Reddit-sized graph:
message() + unsorted_segment_mean (a [114.8 M, K] message tensor) vs the fused rectangular SpMM-mean."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import layers  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
x = torch.randn(n, 128, device=dev, requires_grad=True)
conv = layers.SAGEConv(128, 128, aggr="mean").to(dev)


def ev(fn, reps=3):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for thr, label in ((10**12, "message() + unsorted_segment_mean"), (0, "fused SpMM-mean")):
    layers.FUSED_MIN_EDGES = thr
    f = ev(lambda: conv(x.detach(), ei))
    fb = ev(lambda: conv(x, ei).sum().backward())
    print(f"SAGEConv(128->128, mean) on E={ei.shape[1]}: {label}: fwd {f:.1f} ms, fwd+bwd {fb:.1f} ms  "
          f"(peak {torch.cuda.max_memory_allocated() / 1e9:.0f} GB)", flush=True)
    torch.cuda.reset_peak_memory_stats()
