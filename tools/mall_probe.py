#!/usr/bin/env python3
"""How fast does the K=256 SpMM-sum gather when its source panel fits in the caches?  products-sized row structure
(N_dst = 2 449 029, E = 126 M), sources drawn uniformly from [0, S): S x 1 KiB = the panel the gathers hit.
Tells what a source-blocked SpMM (passes over column blocks, partial sums carried in `out`) could gain."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev).manual_seed(0)
dst = ei[1].contiguous()
w = torch.rand(E, generator=g, device=dev)


def timed(fn, reps=5):
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


print(f"N_dst={n} E={E} K={K}")
for S in (8192, 32768, 65536, 131072, 262144, 524288, 1048576, n):
    src = torch.randint(0, S, (E,), generator=g, device=dev)
    idx = torch.stack([src, dst])
    gp = eng.graph_plan(idx, n, S)
    x = torch.randn(S, K, generator=g, device=dev)
    with torch.no_grad():
        t = timed(lambda: eng.spmm(gp, w, x))
    print(f"  source panel {S:8d} rows = {S * K * 4 / 2**20:7.1f} MiB: {t:7.3f} ms  {E / t / 1e6:6.2f} Gedges/s  "
          f"{(E * (4 * K + 8) + n * (4 * K + 8)) / t / 1e9:6.2f} TB/s algorithmic", flush=True)
    del gp, x, idx, src
    eng.clear_caches()
