#!/usr/bin/env python3
"""north_star: "MFMA tiles only where the feature width makes the SpMM a dense panel".  Is there such a panel?
On the degree-ordered products-sized graph (hubs first: the ordering that concentrates edges the most) count the
edges inside the top-h x top-h source/destination block for h = 2^12, 2^14, 2^16: the block's share of all edges
and its density.  A dense f32-MFMA panel GEMM (x[:h] as the B operand, the block as a dense A) pays when the block
holds a sizeable share of the edges at a density where h*h*K MACs beat gathering nnz rows: MFMA f32 on MI355X
~157 TFLOP/s dense vs 8 TB/s of gathers -> a dense [h, h] x [h, K] product costs 2*h*h*K flops where the sparse walk
costs nnz*(4K + 8) bytes: break-even density = (4K + 8) / (2 h K) * (157e12 / 8e12) ~ 39 / h ... i.e. ~1 % at h = 4096.

    python tools/dense_block_probe.py [products|arxiv|reddit] [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.synth import DATASETS, planted_pairs, rmat_partitioned  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "products"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
n, e, _, _ = DATASETS[name]
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def report(label, src, dst):
    E = int(src.numel())
    say(f"{label}: N={n} E={E}")
    for lg in (10, 12, 14, 16, 18):
        h = 1 << lg
        if h >= n:
            break
        inside = int(((src < h) & (dst < h)).sum())
        rows = int((dst < h).sum())            # edges INTO the top-h rows (any source): what a row-block kernel would own
        say(f"  h=2^{lg:<2d} ({h:7d} nodes = {100.0 * h / n:6.3f} % of N): block edges {inside:11d} = {100.0 * inside / E:6.3f} % of E, "
            f"density {inside / (h * h):.5f}; edges into the top-h rows {100.0 * rows / E:6.2f} % of E")


g = rmat_partitioned(n, e, seed=0, device=dev, relabel="degree")
report(f"{name}-sized R-MAT, degree-ordered (hubs first)", g["src"], g["dst"])
del g
if name == "products":
    s, d = planted_pairs(n, out_deg=max(2, e // (2 * n)), seed=0, device=dev)
    deg = torch.bincount(d, minlength=n)
    rk = torch.empty(n, dtype=torch.int64, device=dev)
    rk[torch.argsort(deg, descending=True, stable=True)] = torch.arange(n, device=dev)
    report("products-sized hierarchical planted graph, degree-ordered", rk[s], rk[d])
    # the other candidate for a dense panel: a community's diagonal block in its natural (class-contiguous) order
    c_hi = (n + 63) // 64
    inside = int(((s < c_hi) & (d < c_hi)).sum())
    say(f"  planted graph, natural order, diagonal block of class 0 ({c_hi} nodes): {inside} edges, density {inside / (c_hi * c_hi):.6f}")
say("verdict: a block qualifies for a dense MFMA panel at >= 10 % of E and >= 1 % density (see the docstring)")
if out_path:
    open(out_path, "w").write("\n".join(lines) + "\n")
