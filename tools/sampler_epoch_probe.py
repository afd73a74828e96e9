#!/usr/bin/env python3
"""The reference's sampling-speed protocol (profiler/sampler/readme.md:9-20: Reddit, one pass over train_idx in
batches of 1024; fan-out [25, 10], and one full-neighbourhood hop [-1]) on the Reddit-sized graph with the device
sampler.  Its published rows (hardware unstated): GGL-CPU 11.26 s / 9.88 s, GGL-GPU 2.28 s / 3.10 s."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.sampler import NeighborSampler  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
n_train = 153_431   # Reddit's train split
train_idx = torch.randperm(n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)[:n_train]
for sizes in ([25, 10], [-1]):
    t0 = time.perf_counter()
    ns = NeighborSampler(ei, sizes, num_nodes=n)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    for rep in range(2):  # second pass = steady state
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        edges = 0
        for b in range(0, n_train, 1024):
            _, n_id, adjs = ns.sample(train_idx[b:b + 1024])
            adjs = adjs if isinstance(adjs, list) else [adjs]
            edges += sum(int(a.edge_index.shape[1]) for a in adjs)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    nb = (n_train + 1023) // 1024
    print(f"sample_lists={sizes}: one pass over {n_train} train nodes in {nb} batches of 1024: {t:.3f} s "
          f"({t / nb * 1e3:.2f} ms/batch, {edges / t / 1e6:.1f} M sampled edges/s); sampler CSR build {t_build:.2f} s", flush=True)
