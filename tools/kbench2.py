#!/usr/bin/env python3
"""tools/kbench2.py — A/B of the long-row chunk size, row ordering and feature-width padding on the
products-sized graph (interleaved rounds, torch.cuda events on the launch stream)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.layers import calc_gcn_norm  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402


def ev_time(fn, reps=3):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS[sys.argv[1] if len(sys.argv) > 1 else "products"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
w = calc_gcn_norm(ei, n).contiguous()
g = torch.Generator(device=dev).manual_seed(0)
xs = {K: torch.randn(n, K, generator=g, device=dev) for K in (256, 64, 48, 47, 16)}
print(f"N={n} E={E}")
for chunk in (4096, 2048, 1024, 512, 256):
    eng.chunk = chunk
    eng.seg_cache.clear(); eng.graph_cache.clear()
    gp = eng.graph_plan(ei, n)
    gp.bwd  # noqa: B018
    for ro in (0, 1, 2):
        eng.set_option("row_order", ro)
        line = f"chunk={chunk:5d} long={gp.fwd.n_long:6d} chunks={gp.fwd.n_chunks:7d} row_order={ro}:"
        for K in (256, 64, 48, 47, 16):
            ms = statistics.median(ev_time(lambda: eng._spmm_fwd("sum", gp.fwd, gp.col, w, xs[K], n)) for _ in range(3))
            msT = statistics.median(ev_time(lambda: eng._spmm_fwd("sum", gp.bwd, gp.colT, w, xs[K], n)) for _ in range(3))
            line += f"  K{K}: {ms:6.2f}/{msT:6.2f}"
        print(line + "  (fwd/bwd ms)", flush=True)
eng.set_option("row_order", 1)
# bias-gradient column sums: own kernel vs torch
for K in (256, 47):
    gg = torch.randn(n, K, generator=g, device=dev)
    print(f"colsum K={K}: ggl {ev_time(lambda: eng.colsum(gg)):.3f} ms   torch.sum(0) {ev_time(lambda: gg.sum(0)):.3f} ms")
