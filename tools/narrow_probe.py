#!/usr/bin/env python3
"""Aggregates at narrow widths and the unfused (message tensor) route: ms, Gedges/s and algorithmic GB/s
(SURVEY.md §8d byte counts).  python tools/narrow_probe.py [arxiv|products]"""
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
g = torch.Generator(device=dev).manual_seed(0)
w = torch.rand(E, generator=g, device=dev)
gp = eng.graph_plan(ei, n)
dst = ei[1].contiguous()
print(f"{name}-sized R-MAT N={n} E={E}; long rows {gp.fwd.n_long}, chunk {gp.fwd.chunk}", flush=True)


def timed(fn, reps=9):
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


for K in (1, 4, 8, 16, 32, 64, 256):
    x = torch.randn(n, K, generator=g, device=dev)
    with torch.no_grad():
        eng.spmm(gp, w, x)
        eng.spmm(gp, w, x)
        t = timed(lambda: eng.spmm(gp, w, x))
    alg = E * (4 * K + 8) + n * (4 * K + 8)
    line = f"K={K:4d} spmm_sum {t:7.3f} ms {E / t / 1e6:7.2f} Gedges/s {alg / t / 1e6:7.0f} GB/s ({alg / t / 8e7:4.0f}% of HBM peak)"
    if E * K * 4 < 60e9:
        msg = torch.randn(E, K, generator=g, device=dev)
        with torch.no_grad():
            for op in ("sum", "max"):
                fn = (lambda: eng.c_segment_sum(msg, dst, n)) if op == "sum" else (lambda: eng.c_segment_max(msg, dst, n))
                ts = timed(fn)
                algs = E * (4 * K + 8) + n * 4 * K + (n * 8 * K if op == "max" else 0)
                line += f" | segment_{op} [E,K] {ts:7.3f} ms {algs / ts / 1e6:6.0f} GB/s ({algs / ts / 8e7:3.0f}%)"
        del msg
    print(line, flush=True)
