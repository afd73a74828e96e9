#!/usr/bin/env python3
"""gspmm mean backward at the products size: the per-edge-divide kernel vs rows pre-divided once + plain SpMM-sum."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
w = torch.rand(ei.shape[1], generator=g, device=dev)


def timed(fn, reps=5):
    for _ in range(2):
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


for K in (256, 128, 48):
    x = torch.randn(n, K, generator=g, device=dev).requires_grad_(True)
    go = torch.randn(n, K, generator=g, device=dev)
    y = eng.c_spmm_mean(ei, w, x)
    res = {}
    for pre in (False, True):
        eng.mean_bwd_prescale = pre
        t = timed(lambda: torch.autograd.grad(y, x, go, retain_graph=True))
        (res[pre],) = torch.autograd.grad(y, x, go, retain_graph=True)
        print(f"K={K} mean backward, prescale={pre}: {t:6.2f} ms", flush=True)
    print("   same bits:", torch.equal(res[False], res[True]))
eng.mean_bwd_prescale = True
