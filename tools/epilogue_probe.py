#!/usr/bin/env python3
"""bias_act forward / backward on [N, K] at the products size: ms and fraction of the HBM roofline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
N = int(os.environ.get("N", "2449029"))


def ev_time(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for K in (256, 48, 47):
    a = torch.randn(N, K, device=dev, requires_grad=True)
    bias = torch.randn(K, device=dev, requires_grad=True)
    g = torch.randn(N, K, device=dev)
    y = eng.bias_act(a, bias, True, 0.5)
    f = ev_time(lambda: eng.bias_act(a, bias, True, 0.5))
    b = ev_time(lambda: torch.autograd.grad(y, (a, bias), g, retain_graph=True))
    gb = N * K * 4 / 1e9
    print(f"K={K}: fwd {f:.3f} ms ({2 * gb / f:.2f} TB/s)  bwd {b:.3f} ms ({3 * gb / b:.2f} TB/s)")
