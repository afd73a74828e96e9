#!/usr/bin/env python3
"""Weight-gradient GEMM gW[out,in] = G^T[out,N] @ X[N,in] with N >> out,in: hipBLASLt's single GEMM vs a
split of the N (reduction) axis into S batched GEMMs + a sum of the S partial products."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)


def ev_time(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def split_wgrad(g, x, S):
    n = g.shape[0]
    m = (n // S) * S
    part = torch.bmm(g[:m].view(S, m // S, -1).transpose(1, 2), x[:m].view(S, m // S, -1)).sum(0)
    if m < n:
        part += g[m:].t() @ x[m:]
    return part


for n in (169343, 2449029):
    for (o, i) in ((256, 256), (256, 128), (40, 256), (256, 100), (47, 256)):
        g = torch.randn(n, o, device=dev)
        x = torch.randn(n, i, device=dev)
        ref = g.t() @ x
        line = f"N={n} out={o} in={i}: mm {ev_time(lambda: g.t() @ x):.3f} ms"
        for S in (8, 32, 128, 512):
            got = split_wgrad(g, x, S)
            err = float((got - ref).abs().max() / ref.abs().max())
            line += f" | S={S} {ev_time(lambda: split_wgrad(g, x, S)):.3f} ms (rel {err:.1e})"
        print(line, flush=True)
