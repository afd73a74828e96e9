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
    tr = GCNTrainer(f, 256, c, num_layers=3, device=dev)  # GCNModel / GCNConv(norm='both') layer classes (norm weights cached per edge_index)
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
    # roofline block of the op's dominant kernel (the destination walk of the forward), bench.py conventions:
    # algorithmic bytes per edge = 4HC (feature row) + 4H (el row) + 4 (col), per output row 4HC + 8H
    import json

    xg = torch.randn(n, H, C, generator=g, device=dev)
    elg, erg = torch.randn(n, H, generator=g, device=dev), torch.randn(n, H, generator=g, device=dev)

    def ev_ms(fn, reps=9):
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
        return statistics.median(ts)

    with torch.no_grad():
        ms_k = ev_ms(lambda: eng.gat_fused(ei, elg, erg, xg, 0.2))
    alg = E * (4 * H * C + 4 * H + 4) + n * (4 * H * C + 8 * H)
    traffic, src = None, None
    pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r2_pmc_gat_reddit.json")
    if os.path.exists(pmc):
        rec = json.load(open(pmc))
        k = [v for kname, v in rec.items() if "gat_fwd2_kernel" in kname]
        if k and "hbm_bytes_per_launch" in k[0]:
            traffic, src = k[0]["hbm_bytes_per_launch"], "profiles/r2_pmc_gat_reddit.json (rocprofv3 --pmc, separate passes)"
    print(json.dumps({"config": 3, "op": f"fused GAT forward {H}x{C}, Reddit-sized graph N={n} E={E}",
                      "roofline": {"bound": "hbm", "kernel": "gat_fwd2_kernel<2,false,true> (+ hub-chunk combine)",
                                   "achieved": alg / ms_k / 1e6, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / ms_k / 1e6 / 8000.0, "traffic": traffic, "traffic_source": src,
                                   "ms_per_launch": ms_k, "alg_bytes_per_launch": alg,
                                   "note": "the 60 MB feature panel is cache-resident: algorithmic bytes above "
                                           "the HBM peak are served by L2 / Infinity Cache, see traffic"}}), flush=True)
    try:
        with torch.no_grad():
            t_u = timeit(lambda: ug(x, ei, n), reps=3, warm=1)
        t_ub = timeit(lambda: fwd_bwd(ug), reps=2, warm=1)
        print(f"    unfused GATConv (segment_softmax + propagate on our segment ops) forward {t_u:.2f} ms, "
              f"forward+backward {t_ub:.2f} ms  -> fused speed-up {t_u / t_f:.1f}x / {t_ub / t_fb:.1f}x")
    except torch.cuda.OutOfMemoryError as ex:
        print("    unfused GATConv: OOM", str(ex)[:80])
    del gp
    # the whole model of config 3: 2-layer, 8-head GAT (602 -> 8 x 8 -> 41 classes, last layer averages its heads),
    # feature and attention dropout 0.6 (examples/gat/gat_trainer.py defaults), one full-graph training step
    from gammagl_amd.layers import GATModel
    import torch.nn.functional as F

    torch.manual_seed(0)
    net = GATModel(602, 8, 41, 8, 0.6, 2, fused=True).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=5e-4)
    yl = torch.randint(0, 41, (n,), generator=g, device=dev)
    tidx = torch.arange(0, n, 3, device=dev)

    def gat_step():
        net.train()
        opt.zero_grad(set_to_none=True)
        F.cross_entropy(net(x, ei, n)[tidx], yl[tidx]).backward()
        opt.step()

    ms = timeit(gat_step, reps=5, warm=2)
    print(f"    2-layer 8-head GATModel(602 -> 8x8 -> 41) training step, dropout 0.6: {ms:.1f} ms "
          f"({2 * 2 * E / ms / 1e6:.2f} Gedges/s over 2 layers x fwd+bwd)", flush=True)
    del net, opt, ei, x
    eng.seg_cache.clear(); eng.graph_cache.clear(); torch.cuda.empty_cache()

if "4" in which:
    from gammagl_amd.sampler import NeighborSampler

    n, e, f, c = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    t0 = time.perf_counter()
    ns = NeighborSampler(ei, [25, 10], num_nodes=n)   # reddit_sage_trainer.py:55-57: sample_lists=[25, 10]
    torch.cuda.synchronize()
    print(f"[4] NeighborSampler CSR build on the products-sized graph: {time.perf_counter() - t0:.2f} s", flush=True)
    x = torch.randn(n, f, generator=g, device=dev)
    sage1 = layers.SAGEConv(f, 256, aggr="mean").to(dev)
    sage2 = layers.SAGEConv(256, c, aggr="mean").to(dev)
    sizes = {}

    def sample_only():
        seeds = torch.randperm(n, generator=g, device=dev)[:2048]
        return ns.sample(seeds)

    def batch():
        _, n_id, adjs = sample_only()
        h = x[n_id]                                       # models/graphsage.py:76-82
        for i, (conv, adj) in enumerate(zip((sage1, sage2), adjs)):
            h = conv((h, h[: adj.size[1]]), adj.edge_index)
            if i == 0:
                h = torch.relu(h)
        h.sum().backward()
        sizes.update(b1=adjs[0].edge_index.shape[1], s1=adjs[0].size, b2=adjs[1].edge_index.shape[1], s2=adjs[1].size)

    batch()
    b0 = eng.stats["plans_built"]
    ms_s = timeit(sample_only, reps=10, warm=2)
    ms = timeit(batch, reps=10, warm=2)
    print(f"    seeds 2048, fanout [25,10]: block1 E={sizes['b1']} {sizes['s1']}, block2 E={sizes['b2']} {sizes['s2']}: "
          f"2-hop sampling {ms_s:.2f} ms, sample + gather + 2x SAGEConv(mean) fwd+bwd {ms:.2f} ms/batch")
    e1n, (ns_, nl) = sizes["b1"], sizes["s1"]
    dstb = torch.sort(torch.randint(0, nl, (e1n,), generator=g, device=dev)).values
    hb = torch.randn(e1n, 256, generator=g, device=dev)
    ms_fresh = timeit(lambda: eng.c_segment_mean(hb, dstb.clone(), nl), reps=20)
    ms_cached = timeit(lambda: eng.c_segment_mean(hb, dstb, nl), reps=20)
    print(f"    unsorted_segment_mean [E={e1n},256] -> [{nl},256]: {ms_cached:.3f} ms with the sampler's CSR plan, "
          f"{ms_fresh:.3f} ms when the plan is rebuilt from the ids (sort + 2 syncs)")
