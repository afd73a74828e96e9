#!/usr/bin/env python3
"""Only the graph-captured GraphSAGE mini-batch step (products-sized graph), for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.sampler import BlockSampler  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402
from gammagl_amd.trainer import SAGEBlockTrainer  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, f, c = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, f, generator=g, device=dev)
y = torch.randint(0, c, (n,), generator=g, device=dev)
bs = BlockSampler(ei, [25, 10], num_nodes=n, eng=eng)
caps = bs.calibrate(2048, trials=8, slack=1.25)
tr = SAGEBlockTrainer(bs, f, 256, c, device=dev, caps=caps)
seeds = torch.randperm(n, generator=g, device=dev)[:2048].contiguous()
tr.capture(x, y, seeds)
torch.cuda.synchronize()
print("REPLAYS START", flush=True)
for _ in range(300):
    seeds.copy_(torch.randint(0, n, (2048,), generator=g, device=dev))
    tr.replay()
torch.cuda.synchronize()
