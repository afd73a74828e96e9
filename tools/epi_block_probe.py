#!/usr/bin/env python3
"""Cost of the fused epilogue on the 64-column-block launches of the K = 256 aggregate (products-sized graph)."""
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
gp = eng.graph_plan(ei, n)
K = 256
x = torch.randn(n, K, generator=g, device=dev)
out = torch.empty(n, K, device=dev)
bias = torch.randn(K, generator=g, device=dev)
rng = eng._rng_state(dev)


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


for cb in (0, 64):
    eng.set_option("col_block", cb)
    t0 = timed(lambda: eng.spmm_sum_into(gp.fwd, gp.col, w, x, out))
    res = [f"plain {t0:6.2f}"]
    for nm, kw in (("bias", dict(bias=bias)), ("bias+relu", dict(bias=bias, relu=True)),
                   ("bias+relu+dropout", dict(bias=bias, relu=True, p_drop=0.5, rng=rng)),
                   ("dropout", dict(p_drop=0.5, rng=rng))):
        t = timed(lambda: eng.spmm_epi_into(gp.fwd, gp.col, w, x, out, epi_K=K, **kw))
        res.append(f"{nm} {t:6.2f}")
    print(f"col_block={cb:3d}: " + " | ".join(res) + " ms", flush=True)
