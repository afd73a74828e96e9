#!/usr/bin/env python3
"""Cost of the halo machinery itself, on ONE MI355X: the products-sized GCN step with the upper half of
the rows treated as remote (world-size-1 RCCL group: send lists, all-to-all-v with itself, halo SpMM in
column chunks, reverse exchange + segment_sum), next to the same step without a halo.  Link time is not
in it (the "remote" rows never leave the GPU), everything else a multi-GPU rank executes is.

    GGL_HALO_CHUNKS=4 python tools/self_halo_probe.py [products|arxiv]
"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.dist import DistGCNTrainer, PartitionedGraph  # noqa: E402
from gammagl_amd.layers import calc_gcn_norm  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "products"
n, e, f, c = DATASETS[name]
ei = rmat_graph(n, e, seed=0, device=dev)
w = calc_gcn_norm(ei, n).contiguous()
E = ei.shape[1]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, f, generator=g, device=dev)
y = torch.randint(0, c, (n,), generator=g, device=dev)
idx = torch.nonzero(torch.rand(n, generator=g, device=dev) < 0.08).reshape(-1)


def run(pg, label, steps=8):
    tr = DistGCNTrainer(pg, f, 256, c, num_layers=3, seed=0, device=dev)
    for _ in range(3):
        tr.step(x, y, idx, idx.numel())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y, idx, idx.numel())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label}: {ms:.1f} ms/step ({tr.net.agg_per_step * E / ms / 1e6:.2f} Gedges/s over {tr.net.agg_per_step} aggregations); halo rows {pg.n_halo}, "
          f"local-source edges {pg.gp_loc.E}, halo-source edges {pg.gp_halo.E if pg.gp_halo else 0}", flush=True)
    return ms


try:
    base = run(PartitionedGraph(ei, w, n, 0, 1, eng=eng), "no halo            ")
    eng.seg_cache.clear(); eng.graph_cache.clear()
    halo = run(PartitionedGraph(ei, w, n, 0, 1, eng=eng, self_halo_from=n // 2), "half the rows 'remote'")
    print(f"halo machinery overhead: {halo - base:.1f} ms/step ({(halo / base - 1) * 100:.0f} %), "
          f"GGL_HALO_CHUNKS={os.environ.get('GGL_HALO_CHUNKS', 'auto')}")
finally:
    dist.destroy_process_group()
