#!/usr/bin/env python3
"""Row hand-out order of the multi-row-per-wave kernels (Engine._row_order): global sort by length vs sort inside
windows of consecutive ids (heavy rows first), on a graph without locality (R-MAT, random / degree ids) and one with
(hierarchical planted communities, random ids vs partition.cluster_order).  K = 256 (4 x 64-column blocks) and K = 64.
    python tools/roworder_probe.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.dist import build_partition  # noqa: E402
from gammagl_amd.synth import DATASETS  # noqa: E402

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


g = torch.Generator(device=dev).manual_seed(0)
for kind, relabel in (("rmat", "random"), ("rmat", "degree"), ("planted", "random"), ("planted", "cluster")):
    eng.row_order_window = 0
    pg = build_partition(n, e, 0, 0, 1, None, dev, eng, relabel=relabel, kind=kind)
    ei, w = pg.ei_loc, pg.w_loc
    say(f"{kind} / {relabel}: E={ei.shape[1]}")
    for K in (256, 64):
        x = torch.randn(n, K, generator=g, device=dev)
        out = torch.empty(n, K, device=dev)
        for win, swz, run in ((0, 0, 0), (2048, 0, 0), (2048, 128, 0), (2048, 0, -1), (2048, 0, 1024), (2048, 0, 4096)):
            eng.clear_caches()
            eng.row_order_window = max(win, 0)
            eng.xcd_run_rows = run      # 0 = off, -1 = decided per graph (GraphPlan._schedule), > 0 = forced run length in rows
            eng.set_option("xcd_swizzle", swz)
            eng.set_option("row_order", 0 if win < 0 else 1)       # -1: no row_order at all (natural id order)
            gp = eng.graph_plan(ei, n)
            eng._sorted_weights(gp.fwd, w)
            t = ev(lambda: eng.spmm_sum_into(gp.fwd, gp.col, w, x, out))
            tt = ev(lambda: eng.spmm_sum_into(gp.bwd, gp.colT, w, x, out))
            label = "natural id order" if win < 0 else ("global sort by length" if win == 0 else f"windows of {win} ids")
            say(f"  K={K:3d} {label:24s} xcd_swizzle={swz:3d} xcd_run_rows={run:5d} (plan: {gp.fwd.xcd_run:4d}, locality "
                f"{gp.locality():.2f}): forward {t:6.2f} ms, transposed {tt:6.2f} ms")
        eng.set_option("xcd_swizzle", 0)
        eng.set_option("row_order", 1)
        eng.xcd_run_rows = -1
        del x, out
    del pg, ei, w
    eng.clear_caches()
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
