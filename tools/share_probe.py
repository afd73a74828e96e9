#!/usr/bin/env python3
"""Config 5 (papers100M-sized, 8 GPUs): ONE rank's real share on ONE MI355X, through the product path —
synth.rmat_partitioned builds the graph piecewise (hash buckets, never a [2, E] tensor) and cuts rank r's
share out of it, dist.PartitionedGraph.from_local builds the halo bookkeeping as a DRY partition (send lists and
buffers exactly as in the 8-rank run, nothing on the wire), and the step runs as the rank would run it minus link
time: send-row gather, local SpMM, halo SpMM with the fused epilogue, the transposed walks, reverse scatter.

    python tools/share_probe.py [papers100M|products] [parts] [rank] [out.json] [relabel: random|degree|none]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.dist import DistGCNTrainer, build_partition  # noqa: E402
from gammagl_amd.synth import DATASETS  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "papers100M"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 3
out_path = sys.argv[4] if len(sys.argv) > 4 else None
relabel = sys.argv[5] if len(sys.argv) > 5 else "random"
n, e, f_in, n_cls = DATASETS[name]
K = 256
res = {"workload": f"{name}-sized R-MAT N={n} E_dir={e}, rank {r} of {P} (dry partition on one GPU), relabel={relabel}"}
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
stats = {}
pg = build_partition(n, e, 0, r, 1, None, dev, eng, parts=P, stats=stats, relabel=relabel)
torch.cuda.synchronize()
res["build_s"] = round(time.perf_counter() - t0, 1)
res["build_peak_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
res.update(e_global=pg.e_global, owned_rows=pg.n_local, local_edges=pg.e_local, local_source_edges=pg.gp_loc.E,
           halo_source_edges=(pg.gp_halo.E if pg.gp_halo else 0), halo_rows=pg.n_halo, send_rows=pg.n_send,
           halo_GB_at_K256=round(pg.n_halo * K * 4 / 1e9, 2), buckets=stats.get("buckets"), rounds=stats.get("rounds"))
print(json.dumps(res), flush=True)
torch.cuda.empty_cache()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(pg.n_local, K, generator=g, device=dev)
bias = torch.zeros(1, K, device=dev)
with torch.no_grad():
    res["aggregate_fwd_ms"] = round(timed(lambda: pg.aggregate(h, bias, relu=True)), 1)
    # the two SpMM blocks on their own (algorithmic bytes SURVEY.md §8d)
    out = torch.empty(pg.n_local, K, device=dev)
    t_loc = timed(lambda: eng.spmm_sum_into(pg.gp_loc.fwd, pg.gp_loc.col, pg.w_loc, h, out))
    recv = torch.randn(pg.n_halo, K, generator=g, device=dev)
    t_halo = timed(lambda: eng.spmm_sum_into(pg.gp_halo.fwd, pg.gp_halo.col, pg.w_halo, recv, out, accumulate=True))
    alg = lambda E: E * (4 * K + 8) + pg.n_local * (4 * K + 8)  # noqa: E731
    res["spmm_local_ms"], res["spmm_local_GBs"] = round(t_loc, 1), round(alg(pg.gp_loc.E) / t_loc / 1e6)
    res["spmm_halo_ms"], res["spmm_halo_GBs"] = round(t_halo, 1), round(alg(pg.gp_halo.E) / t_halo / 1e6)
    del out, recv
hg = h.clone().requires_grad_(True)
go = torch.randn(pg.n_local, K, generator=g, device=dev)


def fb():
    hg.grad = None
    pg.aggregate(hg, bias, relu=True).backward(go)


res["aggregate_fwd_bwd_ms"] = round(timed(fb), 1)
res["edges_per_s_fwd_bwd"] = round(2 * pg.e_local / res["aggregate_fwd_bwd_ms"] * 1e3)
# column checksum of the local-source block in f64 on a column sample (property test at full size)
with torch.no_grad():
    y = torch.empty(pg.n_local, K, device=dev)
    eng.spmm_sum_into(pg.gp_loc.fwd, pg.gp_loc.col, pg.w_loc, h, y)
    chk = torch.zeros(8, dtype=torch.float64, device=dev)
    src = pg.ei_loc[0]
    for s in range(0, src.numel(), 32_000_000):
        sl = slice(s, min(src.numel(), s + 32_000_000))
        chk += (pg.w_loc[sl].double().unsqueeze(1) * h[src[sl], :8].double()).sum(0)
    res["checksum_rel_err"] = float(((y[:, :8].double().sum(0) - chk).abs() / chk.abs().clamp(min=1e-9)).max())
    del y
del hg, go, h
torch.cuda.empty_cache()
res["aggregate_peak_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
print(json.dumps(res), flush=True)
# one full training step of the share (3-layer GCN, hidden 256), as the rank would run it minus link time
try:
    x = torch.randn(pg.n_local, f_in, generator=g, device=dev)
    yl = torch.randint(0, n_cls, (pg.n_local,), generator=g, device=dev)
    tl = torch.nonzero(torch.rand(pg.n_local, generator=g, device=dev) < 0.01).reshape(-1)
    tr = DistGCNTrainer(pg, f_in, K, n_cls, num_layers=3, seed=0, device=dev)
    ms = timed(lambda: tr.step(x, yl, tl, tl.numel() * P), reps=2)
    res["train_step_ms"] = round(ms, 1)
    res["train_step_edges_per_s_x8"] = round(tr.net.agg_per_step * pg.e_global / ms * 1e3)
    res["train_step_peak_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
except torch.OutOfMemoryError as ex:  # noqa: PERF203
    res["train_step_ms"] = None
    res["train_step_note"] = "out of memory on one GPU: " + str(ex)[:120]
print(json.dumps(res), flush=True)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
