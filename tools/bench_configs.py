#!/usr/bin/env python3
"""tools/bench_configs.py — the secondary BASELINE.json configs on one MI355X (kernel-level numbers for
DESIGN.md; the graded metric is bench.py):

  [2] 3-layer GCN hidden=256, ogbn-arxiv-sized graph           (CSR SpMM + unsorted_segment_sum path)
  [3] 8-head GAT layer, Reddit-sized graph                     (fused edge-softmax + aggregate vs unfused)
  [4] GraphSAGE neighbour-sampled blocks [25,10], batch 2048, products-sized graph (segment_mean on
      rectangular blocks; a NEW edge list per batch, so the plan build is paid every time)
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine, layers  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402
from gammagl_amd.trainer import GCNTrainer  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
g = torch.Generator(device=dev).manual_seed(0)
which = sys.argv[1:] or ["2", "3", "4"]


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


if "2" in which:
    n, e, f, c = DATASETS["arxiv"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    E = ei.shape[1]
    x = torch.randn(n, f, generator=g, device=dev)
    y = torch.randint(0, c, (n,), generator=g, device=dev)
    idx = torch.randperm(n, generator=g, device=dev)[: n // 2]
    tr = GCNTrainer(f, 256, c, num_layers=3, device=dev)  # faithful GCNConv: degrees recomputed per layer
    ms = timeit(lambda: tr.step(x, ei, y, idx, n), reps=10, warm=3)
    print(f"[2] arxiv-sized 3-layer GCN h=256 train step (GCNConv norm='both', fused gspmm route): {ms:.2f} ms "
          f"-> {6 * E / ms / 1e6:.2f} Gedges/s", flush=True)
    msg = torch.randn(E, 256, generator=g, device=dev)
    dst = ei[1].contiguous()
    for nm, fn in (("sum", eng.c_segment_sum), ("mean", eng.c_segment_mean), ("max", eng.c_segment_max)):
        ms = timeit(lambda: fn(msg, dst, n), reps=10)
        print(f"    unsorted_segment_{nm} [E={E},256] -> [N,256]: {ms:.3f} ms  ({E / ms / 1e6:.2f} Gedges/s)")
    del msg

if "3" in which:
    n, e, _, _ = DATASETS["reddit"]
    t0 = time.perf_counter()
    ei = rmat_graph(n, e, seed=0, device=dev)
    torch.cuda.synchronize()
    E = ei.shape[1]
    print(f"[3] reddit-sized graph N={n} E={E} generated in {time.perf_counter() - t0:.1f}s", flush=True)
    H, C = 8, 8
    fg = layers.FusedGATConv(602, C, heads=H).to(dev)
    ug = layers.GATConv(602, C, heads=H).to(dev)
    ug.load_state_dict(fg.state_dict())
    x = torch.randn(n, 602, generator=g, device=dev)
    gp = eng.graph_plan(ei, n)
    gp.bwd, gp.posT  # noqa: B018
    print(f"    max in-degree {int(gp.fwd.counts().max())}")

    def fwd_bwd(layer):
        xx = x.clone().requires_grad_(True)
        out = layer(xx, ei, n)
        out.sum().backward()

    with torch.no_grad():
        t_f = timeit(lambda: fg(x, ei, n), reps=5)
    t_fb = timeit(lambda: fwd_bwd(fg), reps=3, warm=1)
    print(f"    FusedGATConv 8x8 forward {t_f:.2f} ms ({E / t_f / 1e6:.2f} Gedges/s), forward+backward {t_fb:.2f} ms")
    try:
        with torch.no_grad():
            t_u = timeit(lambda: ug(x, ei, n), reps=3, warm=1)
        t_ub = timeit(lambda: fwd_bwd(ug), reps=2, warm=1)
        print(f"    unfused GATConv (segment_softmax + propagate on our segment ops) forward {t_u:.2f} ms, "
              f"forward+backward {t_ub:.2f} ms  -> fused speed-up {t_u / t_f:.1f}x / {t_ub / t_fb:.1f}x")
    except torch.cuda.OutOfMemoryError as ex:
        print("    unfused GATConv: OOM", str(ex)[:80])
    del ei, gp, x
    eng.seg_cache.clear(); eng.graph_cache.clear(); torch.cuda.empty_cache()

if "4" in which:
    n, e, f, c = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    plan = eng.seg_plan(ei[1].contiguous(), n)          # CSR by destination = in-neighbour lists
    rowptr, col = plan.rowptr, eng.gather_i32(ei[0].contiguous(), plan.perm).long()
    x = torch.randn(n, f, generator=g, device=dev)

    def sample_block(seeds, fanout):
        deg = rowptr[seeds + 1] - rowptr[seeds]
        k = torch.clamp(deg, max=fanout)
        owner = torch.repeat_interleave(torch.arange(seeds.numel(), device=dev), k)
        start = torch.cumsum(k, 0) - k
        pos_in = torch.arange(owner.numel(), device=dev) - start[owner]
        r = torch.rand(owner.numel(), generator=g, device=dev)
        off = torch.where(deg[owner] <= fanout, pos_in, (r * deg[owner]).long())
        src = col[rowptr[seeds][owner] + off]
        n_id, inv = torch.unique(torch.cat([seeds, src]), return_inverse=True)  # dst nodes first? keep both maps
        return src, owner, n_id, inv

    sage1 = layers.SAGEConv(f, 256, aggr="mean").to(dev)
    sage2 = layers.SAGEConv(256, c, aggr="mean").to(dev)

    def batch():
        seeds = torch.randint(0, n, (2048,), generator=g, device=dev)
        s2, o2, _, _ = sample_block(seeds, 10)                      # hop 2 (outer): seeds <- 10 neighbours
        l1 = torch.unique(torch.cat([seeds, s2]))                   # nodes of layer 1
        s1, o1, _, _ = sample_block(l1, 25)                         # hop 1: layer-1 nodes <- 25 neighbours
        src_nodes, inv = torch.unique(torch.cat([l1, s1]), return_inverse=True)
        # block 1: N_src = |src_nodes| -> N_dst = |l1|
        e1 = torch.stack([inv[l1.numel():], torch.searchsorted(l1, l1[o1])])
        h0 = x[src_nodes]
        dst_feat = x[l1]
        h1 = torch.relu(sage1((h0, dst_feat), e1))
        e2 = torch.stack([torch.searchsorted(l1, s2), o2])
        pos_seed = torch.searchsorted(l1, seeds)
        out = sage2((h1, h1[pos_seed]), e2)
        out.sum().backward()
        return e1.shape[1], e2.shape[1], src_nodes.numel(), l1.numel()

    e1n, e2n, ns, nl = batch()
    b0 = eng.stats["plans_built"]
    ms = timeit(batch, reps=10, warm=2)
    print(f"[4] SAGE mini-batch (seeds 2048, fanout [25,10]) block1 E={e1n} ({ns}->{nl}), block2 E={e2n} ({nl}->2048): "
          f"sample + 2x SAGEConv(mean) fwd+bwd = {ms:.2f} ms/batch; plans built per batch = "
          f"{(eng.stats['plans_built'] - b0) / 12:.1f}")
    # the aggregate alone on a block-1 sized problem, plan build included (new edge list every batch)
    srcb = torch.randint(0, ns, (e1n,), generator=g, device=dev)
    dstb = torch.sort(torch.randint(0, nl, (e1n,), generator=g, device=dev)).values
    hb = torch.randn(e1n, 256, generator=g, device=dev)

    def agg_fresh():
        d = dstb.clone()  # a new tensor = a new plan
        return eng.c_segment_mean(hb, d, nl)

    ms_fresh = timeit(agg_fresh, reps=20)
    ms_cached = timeit(lambda: eng.c_segment_mean(hb, dstb, nl), reps=20)
    print(f"    unsorted_segment_mean [E={e1n},256] -> [{nl},256]: {ms_cached:.3f} ms cached plan, {ms_fresh:.3f} ms incl. plan build")
