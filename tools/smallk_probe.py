#!/usr/bin/env python3
"""segment_sum over [E, K] for small K on the Reddit-sized graph vs the long-row threshold."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS[sys.argv[1] if len(sys.argv) > 1 else "reddit"]
ei = rmat_graph(n, e, seed=0, device=dev, order="dst")
E = ei.shape[1]
dst = ei[1].contiguous()


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


print(f"N={n} E={E} (ids sorted: no perm)")
for K in (1, 8, 16, 64):
    x = torch.randn(E, K, device=dev)
    line = f"K={K:3d}:"
    for chunk in (4096, 1024, 256, 64):
        eng.chunk = chunk
        eng.seg_cache.clear()
        plan = eng.seg_plan(dst, n)
        ms = ev_time(lambda: eng._segment_fwd("sum", x, plan))
        line += f"  chunk={chunk}: {ms:.3f} ms ({E * (4 * K + 8) / ms / 1e9:.2f} TB/s)"
    print(line, flush=True)
    del x

# narrow-row unroll A/B at the default threshold
eng.chunk = 0
eng.seg_cache.clear()
plan = eng.seg_plan(dst, n)
for K in (1, 8, 16):
    x = torch.randn(E, K, device=dev)
    line = f"K={K:3d} (chunk auto={plan.chunk}):"
    for u in (4, 16):
        eng.set_option("unroll_narrow", u)
        ms = ev_time(lambda: eng._segment_fwd("sum", x, plan))
        msx = ev_time(lambda: eng._segment_fwd("max", x, plan))
        line += f"  U={u}: sum {ms:.3f} ms ({E * (4 * K + 8) / ms / 1e9:.2f} TB/s), max {msx:.3f} ms"
    print(line, flush=True)
    del x
eng.set_option("unroll_narrow", 16)
