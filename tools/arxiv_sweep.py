#!/usr/bin/env python3
"""Chunk-size / unroll sweep of the K=256 SpMM-sum and segment_sum on the arxiv-sized graph (small E:
the launch is a handful of waves deep, so the longest unsplit row is the critical path)."""
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
x = torch.randn(n, 256, generator=g, device=dev)
w = torch.rand(E, generator=g, device=dev)
msg = torch.randn(E, 256, generator=g, device=dev) if E < 20_000_000 else None


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


print(f"{name}: N={n} E={E}")
for unroll in (4, 8):
    eng.set_option("unroll", unroll)
    for chunk in (4096, 2048, 1024, 512, 256, 128):
        eng.chunk = chunk
        gp = eng.graph_plan(ei.clone(), n)
        ms = eng.time_spmm_sum(gp, w, x, reps=20)
        alg = E * 1032 + n * 1032
        line = f"unroll={unroll} chunk={chunk:5d} long={gp.fwd.n_long:6d} chunks={gp.fwd.n_chunks:7d} spmm K256 {ms:.3f} ms ({alg / ms / 1e9:.2f} TB/s alg)"
        if msg is not None:
            ids = ei[1].clone()
            ms2 = ev_time(lambda: eng.c_segment_sum(msg, ids, n))
            line += f"  segment_sum [E,256] {ms2:.3f} ms"
        print(line, flush=True)
eng.set_option("unroll", 4)
