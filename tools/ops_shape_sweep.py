#!/usr/bin/env python3
"""Cliff hunt over the op surface on the arxiv- and products-sized graphs: every op x dtype x width combination a
layer can produce, ms and algorithmic TB/s; anything far below its neighbours is a shape cliff."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
n, e, _, _ = DATASETS[name]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
dst = ei[1].contiguous()
w = torch.rand(E, device=dev)


def ev(fn, reps=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print(f"{name}: N={n} E={E}")
for dt, esz in ((torch.float32, 4), (torch.float16, 2), (torch.bfloat16, 2), (torch.float64, 8), (torch.int32, 4), (torch.int64, 8)):
    for K in (1, 7, 16, 47, 64, 256):
        if E * K * esz > 60e9:
            continue
        x = (torch.randn(E, K, device=dev) * 4).to(dt)
        line = f"segment {str(dt)[6:]:9s} K={K:3d}:"
        for op in ("sum", "mean", "max"):
            if op == "mean" and dt in (torch.int32, torch.int64) and False:
                continue
            fn = {"sum": eng.c_segment_sum, "mean": eng.c_segment_mean, "max": eng.c_segment_max}[op]
            ms = ev(lambda: fn(x, dst, n))
            line += f"  {op} {ms:7.3f} ms ({E * (K * esz + 8) / ms / 1e9:5.2f} TB/s)"
        print(line, flush=True)
        del x
for K in (7, 16, 41, 47, 100, 256, 602):
    x = torch.randn(n, K, device=dev)
    line = f"gspmm f32 K={K:3d}:"
    for op, fn in (("sum", eng.c_spmm_sum), ("mean", eng.c_spmm_mean), ("max", eng.c_spmm_max)):
        ms = ev(lambda: fn(ei, w, x))
        line += f"  {op} {ms:7.3f} ms ({E * (4 * K + 8) / ms / 1e9:5.2f} TB/s)"
    print(line, flush=True)
for H, C in ((8, 8), (8, 41), (4, 7), (1, 256), (16, 16)):
    x = torch.randn(n, H, C, device=dev, requires_grad=True)
    wh = torch.rand(E, H, device=dev, requires_grad=True)
    f = ev(lambda: eng.c_bspmm_sum(ei, wh.detach(), x.detach()))
    go = torch.randn(n, H, C, device=dev)

    def fwd_bwd():
        # gradients dropped first: AccumulateGrad on an existing [E, H] gradient is a 12 GB read-modify-write of its own
        # (rounds 3-4 timed it as part of "fwd+bwd": 16.4 ms instead of 13.x at 8 x 8)
        x.grad = None
        wh.grad = None
        eng.c_bspmm_sum(ei, wh, x).backward(go)

    fb = ev(fwd_bwd)
    print(f"bspmm H={H:2d} C={C:3d}: fwd {f:7.3f} ms ({E * (4 * H * C + 4 * H + 4) / f / 1e9:5.2f} TB/s)  fwd+bwd {fb:7.3f} ms "
          f"({fb / f:4.2f}x fwd)", flush=True)
