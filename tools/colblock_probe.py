#!/usr/bin/env python3
"""K = 256 SpMM-sum on the products-sized graph as ONE launch vs as 2 / 4 / 8 launches over column blocks of the same
[N, 256] matrices (row strides passed down: no copies).  Narrower slices keep more distinct hub rows in L2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "products"
n, e, _, _ = DATASETS[name]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
g = torch.Generator(device=dev).manual_seed(0)
w = torch.rand(E, generator=g, device=dev)
gp = eng.graph_plan(ei, n)
gp.bwd  # noqa: B018


def timed(fn, reps=7):
    for _ in range(3):
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


print(f"{name}: N={n} E={E}")
for K in (256, 128, 96, 100):
    x = torch.randn(n, K, generator=g, device=dev)
    out = torch.empty(n, K, device=dev)
    bias = torch.randn(K, generator=g, device=dev)
    ref = None
    for nb in ((1, 2, 4, 8) if K >= 128 else (1, 2)):
        wd = -(-K // nb // 4) * 4 if nb > 1 else K

        def run(plan=gp.fwd, col=gp.col):
            for b in range(nb):
                c0, c1 = b * wd, min(K, (b + 1) * wd)
                eng.spmm_sum_into(plan, col, w, x[:, c0:c1], out[:, c0:c1])

        t = timed(run)
        run()
        if ref is None:
            ref = out.clone()
        same = torch.equal(ref, out)
        tb = timed(lambda: run(gp.bwd, gp.colT))
        print(f"  K={K} in {nb} column block(s) of {wd}: forward {t:6.3f} ms, transposed {tb:6.3f} ms, same bits: {same}", flush=True)
