#!/usr/bin/env python3
"""Config 4: one GraphSAGE mini-batch step (2048 seeds, fan-out [25, 10], hidden 256) on the products-sized
graph — the dynamic NeighborSampler path (two host reads per hop) vs the static-shape BlockSampler eager and
as ONE replayed hipGraph.  python tools/sage_minibatch_probe.py [batches]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.sampler import BlockSampler, NeighborSampler  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402
from gammagl_amd.trainer import SAGEBlockTrainer, SAGETrainer  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, f, c = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, f, generator=g, device=dev)
y = torch.randint(0, c, (n,), generator=g, device=dev)
B, reps = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 50
batches = [torch.randperm(n, generator=g, device=dev)[:B].contiguous() for _ in range(reps)]


def wall(fn, label):
    for b in batches[:5]:
        fn(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        fn(b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(batches) * 1e3
    print(f"{label}: {ms:.2f} ms/batch", flush=True)
    return ms


dyn = SAGETrainer(NeighborSampler(ei, [25, 10], num_nodes=n, eng=eng), f, 256, c, device=dev)
wall(lambda b: dyn.step(x, y, b), "dynamic sampler, eager        ")
bs = BlockSampler(ei, [25, 10], num_nodes=n, eng=eng)
print("block capacities (dst, src, edges), innermost first:", bs.capacities(B))
caps = bs.calibrate(B, trials=8, slack=1.25)
print("calibrated capacities (src, edges), innermost first:", caps, flush=True)
blk = SAGEBlockTrainer(bs, f, 256, c, device=dev, caps=caps)
wall(lambda b: blk.step(x, y, b), "static-shape sampler, eager   ")
seeds = batches[0].clone()
blk.capture(x, y, seeds)


def rep(b):
    seeds.copy_(b)
    blk.replay()


wall(rep, "static-shape sampler, hipGraph")
print("hops that overflowed their capacity so far:", bs.overflow_count())
n_id, blocks, counts = bs.sample(batches[0], caps=caps)
print("valid / capacity:", [(int(b.counts[0]), b.n_src_cap, int(b.counts[1]), b.e_cap) for b in blocks])
# sampler alone
for lbl, fn in (("dynamic sample()", lambda b: dyn.sampler.sample(b)), ("static  sample()", lambda b: bs.sample(b, caps=caps))):
    wall(fn, lbl + "               ")
