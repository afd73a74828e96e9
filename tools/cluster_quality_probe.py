#!/usr/bin/env python3
"""How much of a planted hierarchy partition.cluster_order recovers from randomly labelled nodes at the products size,
and what that is worth to the aggregate: purity of the found communities against the planted fine / mid groups, share
of the edges whose endpoints end up within one fine / one mid group's width of each other, K = 256 aggregate time —
for a few settings of the propagation, beside the random order and the generator's own.
    python tools/cluster_quality_probe.py [out.txt]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.partition import cluster_order, relabel_edges  # noqa: E402
from gammagl_amd.synth import DATASETS, PLANTED_LEVELS, planted_pairs  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["products"]
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def ev(fn, reps=5, warm=2):
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


s_, d_ = planted_pairs(n, out_deg=max(2, e // (2 * n)), seed=0, device=dev)
pi = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(dev)
inv = torch.empty_like(pi)
inv[pi] = torch.arange(n, device=dev)
ei = torch.stack([pi[s_], pi[d_]]).contiguous()
nat = torch.stack([s_, d_]).contiguous()
G_f, G_m = PLANTED_LEVELS[0][0], PLANTED_LEVELS[1][0]
x = torch.randn(n, 256, device=dev)
out = torch.empty(n, 256, device=dev)


def purity(lab, G):
    true = (inv * G) // n
    cnt = torch.bincount(lab * G + true, minlength=(int(lab.max()) + 1) * G).view(-1, G)
    return float(cnt.max(1).values.sum()) / n


def near(e2, W):
    return float(((e2[0] - e2[1]).abs() < W).float().mean())


def agg_ms(edges):
    eng.clear_caches()
    gp = eng.graph_plan(edges, n)
    t = ev(lambda: eng.spmm_sum_into(gp.fwd, gp.col, None, x, out))
    return t, gp.fwd.xcd_run, gp.locality()


say(f"hierarchical planted graph N={n} E={ei.shape[1]}, levels {PLANTED_LEVELS}; fine group = {n // G_f} nodes, mid group = {n // G_m}")
from gammagl_amd.partition import halo_stats as _hs  # noqa: E402

for label, edges in (("random ids", ei), ("generator's own order", nat)):
    t, run, loc = agg_ms(edges)
    h8, share8 = _hs(edges, n, 8)
    say(f"  {label:42s}: edges within a fine / mid width {near(edges, n // G_f):.3f} / {near(edges, n // G_m):.3f}; K=256 aggregate {t:6.2f} ms "
        f"(xcd_run {run}, locality {loc:.2f}); P=8: halo rows {h8}, local-source share {share8:.2f}")
from gammagl_amd.partition import halo_stats  # noqa: E402

for C, sw, up in ((1020, 20, 1.0), (4080, 20, 1.0), (8160, 20, 1.0), (4080, 30, 1.0)):
    eng.clear_caches()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rk, lab = cluster_order(ei, n, clusters=C, sweeps=sw, seed=0, eng=eng, update=up)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e2 = relabel_edges(ei, rk).contiguous()
    t, run, loc = agg_ms(e2)
    h8, share8 = halo_stats(e2, n, 8)
    say(f"  cluster_order({C:4d} labels, {sw} sweeps) {dt:5.1f} s, {int(torch.unique(lab).numel())} communities left: purity fine / mid "
        f"{purity(lab, G_f):.3f} / {purity(lab, G_m):.3f}; edges within a fine / mid width {near(e2, n // G_f):.3f} / {near(e2, n // G_m):.3f}; "
        f"K=256 aggregate {t:6.2f} ms (xcd_run {run}, locality {loc:.2f}); P=8: halo rows {h8}, local-source share {share8:.2f}")
    del rk, lab, e2
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
