#!/usr/bin/env python3
"""north_star's "MFMA tiles only where ... the SpMM is a dense panel", prototyped: on the DEGREE-ordered
products-sized graph the top-h x top-h block is dense (tools/dense_block_probe.py: h = 16384 holds 16 % of the edges
at 7.7 % density).  Split the aggregate: the block as a dense f32 matrix through an MFMA GEMM
(hipBLASLt v_mfma_f32_*_f32: exact f32 products, 157 TFLOP/s peak) added onto the sparse walk over the remaining
edges — vs the one sparse walk over everything.  K = 256 and 64.
    python tools/dense_block_mfma_probe.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_partitioned  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["products"]
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def ev(fn, reps=7, warm=2):
    for _ in range(warm):
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


gph = rmat_partitioned(n, e, seed=0, device=dev, relabel="degree")
src, dst, w = gph["src"], gph["dst"], gph["w"]
E = int(src.numel())
ei = torch.stack([src, dst]).contiguous()
gp = eng.graph_plan(ei, n)
eng._sorted_weights(gp.fwd, w)
g = torch.Generator(device=dev).manual_seed(0)
say(f"degree-ordered products-sized R-MAT N={n} E={E}; f32 throughout")
for K in (256, 64):
    x = torch.randn(n, K, generator=g, device=dev)
    full = torch.empty(n, K, device=dev)
    t_full = ev(lambda: eng.spmm_sum_into(gp.fwd, gp.col, w, x, full))
    say(f"K={K}: one sparse walk over all edges: {t_full:.2f} ms")
    for h in (4096, 8192, 16384, 32768):
        inb = (src < h) & (dst < h)
        nb = int(inb.sum())
        a = torch.zeros(h, h, device=dev)
        a.index_put_((dst[inb], src[inb]), w[inb], accumulate=True)
        ei_r = ei[:, ~inb].contiguous()
        w_r = w[~inb].contiguous()
        gp_r = eng.graph_plan(ei_r, n)
        eng._sorted_weights(gp_r.fwd, w_r)
        out = torch.empty(n, K, device=dev)

        def split():
            eng.spmm_sum_into(gp_r.fwd, gp_r.col, w_r, x, out)
            out[:h].addmm_(a, x[:h])

        t_split = ev(split)
        t_rest = ev(lambda: eng.spmm_sum_into(gp_r.fwd, gp_r.col, w_r, x, out))
        t_gemm = ev(lambda: torch.mm(a, x[:h]))
        split()
        err = float((out - full).abs().max() / full.abs().max())
        say(f"   h={h:6d}: block {nb:9d} edges = {100.0 * nb / E:5.2f} % of E, density {nb / h / h:.4f}, dense A {h * h * 4 / 2**30:5.2f} GiB | "
            f"sparse rest {t_rest:6.2f} ms + MFMA GEMM [{h},{h}]x[{h},{K}] {t_gemm:5.2f} ms ({2.0 * h * h * K / t_gemm / 1e9:6.1f} TFLOP/s) "
            f"= split {t_split:6.2f} ms vs {t_full:6.2f} ms ({100.0 * (t_full - t_split) / t_full:+5.1f} %), max |diff| / max |y| = {err:.1e}")
        del a, ei_r, w_r, gp_r, out, inb
        eng.graph_cache.d.pop(next(reversed(eng.graph_cache.d)), None)
        torch.cuda.empty_cache()
    del x, full
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
