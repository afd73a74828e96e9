#!/usr/bin/env python3
"""bspmm forward / forward+backward on the products-sized graph, A/B of the weight gradient: the thread-per-item /
ripple kernels in edge order (backward.hip) vs the walk along the sorted plan with LDS-staged strips (edgedot.hip).
    python tools/bspmm_bwd_probe.py [out.txt]"""
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
E = ei.shape[1]
g = torch.Generator(device=dev).manual_seed(0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def ev(fn, reps=5, warm=2):
    for _ in range(warm):
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


say(f"products-sized R-MAT N={n} E={E}; bspmm_sum forward, forward+backward (gx + gw), and the weight gradient alone")
for (H, C) in ((1, 256), (8, 44), (8, 32), (4, 64), (1, 64), (8, 8), (2, 128)):
    x = torch.randn(n, H, C, generator=g, device=dev).requires_grad_(True)
    w = torch.rand(E, H, generator=g, device=dev).requires_grad_(True)
    go = torch.randn(n, H, C, generator=g, device=dev)
    with torch.no_grad():
        t_f = ev(lambda: eng.c_bspmm_sum(ei, w, x))
    res = {}
    for mode in (False, True):
        eng.gradw_sorted = mode

        def fb():
            x.grad = w.grad = None
            eng.c_bspmm_sum(ei, w, x).backward(go)

        t_fb = ev(fb, reps=3, warm=1)
        res[mode] = (t_fb, w.grad.clone())
    same = torch.equal(res[False][1], res[True][1])
    say(f"  H={H:2d} C={C:3d} (K={H * C:3d}): fwd {t_f:7.2f} ms | fwd+bwd edge-order gw {res[False][0]:7.2f} ms ({res[False][0] / t_f:4.1f}x fwd) "
        f"| sorted-plan gw {res[True][0]:7.2f} ms ({res[True][0] / t_f:4.1f}x fwd) | gw bit-identical: {same}")
    del x, w, go, res
    torch.cuda.empty_cache()
eng.gradw_sorted = True
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
