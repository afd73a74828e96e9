#!/usr/bin/env python3
"""Fused GAT on the Reddit-sized graph: fast kernels vs the generic ones, forward and forward+backward,
with and without attention dropout (hipEvent timing, median of reps).  python tools/gat_probe2.py [H C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
H, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 8)
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, H, C, generator=g, device=dev)
el, er = torch.randn(n, H, generator=g, device=dev), torch.randn(n, H, generator=g, device=dev)
go = torch.randn(n, H, C, generator=g, device=dev)
gp = eng.graph_plan(ei, n)
gp.bwd, gp.posT  # noqa: B018  (plans built outside the timed region)
print(f"Reddit-sized R-MAT N={n} E={E} H={H} C={C}: long rows {gp.fwd.n_long} ({gp.fwd.n_chunks} chunks of {gp.fwd.chunk}), "
      f"max row {gp.fwd.max_len}; fast path supported: {bool(eng.lib.ggl_gat_fast_supported(H, C))}", flush=True)


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


alg = E * (4 * H * C + 4 * H + 4) + n * (4 * H * C + 8 * H)
res = {}
for fast in (True, False):
    eng.gat_fast = fast
    for p in (0.0, 0.6):
        def fwd():
            with torch.no_grad():
                return eng.gat_fused(ei, el, er, x, 0.2, dropout_rate=p)

        def fwdbwd():
            xa, ea, eb = (t.detach().requires_grad_(True) for t in (x, el, er))
            eng.gat_fused(ei, ea, eb, xa, 0.2, dropout_rate=p).backward(go)

        tf, tb = timed(fwd), timed(fwdbwd)
        res[(fast, p)] = (tf, tb)
        print(f"{'fast   ' if fast else 'generic'} p_drop={p}: forward {tf:.2f} ms ({E / tf / 1e6:.1f} Gedges/s, "
              f"{alg / tf / 1e6:.0f} GB/s algorithmic), forward+backward {tb:.2f} ms", flush=True)
eng.gat_fast = True
xa, ea, eb = (t.detach().requires_grad_(True) for t in (x, el, er))
ya = eng.gat_fused(ei, ea, eb, xa, 0.2)
ya.backward(go)
eng.gat_fast = False
xb, ec, ed = (t.detach().requires_grad_(True) for t in (x, el, er))
yb = eng.gat_fused(ei, ec, ed, xb, 0.2)
yb.backward(go)
for nm, a, b in (("out", ya, yb), ("gx", xa.grad, xb.grad), ("gel", ea.grad, ec.grad), ("ger", eb.grad, ed.grad)):
    print(f"  fast vs generic {nm}: max abs diff {float((a - b).abs().max()):.3e} (max |generic| {float(b.abs().max()):.3e})")
