#!/usr/bin/env python3
"""f32 rows that are not made of aligned float4s (K % 4 != 0): the ragged 4-floats-per-lane kernels against the
VEC = 1 kernels they replace (ggl_set_option("ragged4", 0)), same bits required.  products- or arxiv-sized graph."""
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
dst = ei[1].contiguous()


def timed(fn, reps=7):
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


print(f"{name}: N={n} E={E}")
for K in (5, 7, 10, 41, 47, 101, 258):
    x = torch.randn(n, K, generator=g, device=dev)
    line = f"K={K:4d}"
    for nm, fn in (("gspmm sum", lambda: eng.c_spmm_sum(ei, w, x)), ("gspmm max", lambda: eng.c_spmm_max(ei, w, x))):
        with torch.no_grad():
            eng.set_option("ragged4", 1)
            t1, y1 = timed(fn), fn()
            eng.set_option("ragged4", 0)
            t0, y0 = timed(fn), fn()
            eng.set_option("ragged4", 1)
        alg = E * (4 * K + 8) + n * (4 * K + 8)
        line += f"  {nm}: {t1:7.3f} ms ({alg / t1 / 1e9:5.2f} TB/s) vs VEC=1 {t0:7.3f} ms, {'same bits' if torch.equal(y1, y0) else 'DIFFERENT'}"
    if K <= 64 and name != "products" or K <= 47:
        msg = torch.randn(E, K, generator=g, device=dev)
        with torch.no_grad():
            f = lambda: eng.c_segment_sum(msg, dst, n)  # noqa: E731
            eng.set_option("ragged4", 1)
            t1, y1 = timed(f), f()
            eng.set_option("ragged4", 0)
            t0, y0 = timed(f), f()
            eng.set_option("ragged4", 1)
        line += f"  segment_sum [E,K]: {t1:7.3f} vs {t0:7.3f} ms, {'same bits' if torch.equal(y1, y0) else 'DIFFERENT'}"
        del msg
    print(line, flush=True)
