#!/usr/bin/env python3
"""arxiv-sized 3-layer GCN step (config 2): eager vs the whole step captured into one hipGraph (trainer.GraphedStep)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.dist import DistGCNTrainer, build_partition  # noqa: E402
from gammagl_amd.synth import DATASETS  # noqa: E402
from gammagl_amd.trainer import GraphedStep  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
name = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
n, e, f, c = DATASETS[name]
pg = build_partition(n, e, 0, 0, 1, None, dev, eng)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, f, generator=g, device=dev)
y = torch.randint(0, c, (n,), generator=g, device=dev)
idx = torch.nonzero(torch.rand(n, generator=g, device=dev) < 0.08).reshape(-1)
tr = DistGCNTrainer(pg, f, 256, c, num_layers=3, seed=0, device=dev, capturable=True)


def wall(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t_e = wall(lambda: tr.step(x, y, idx, idx.numel()))
print(f"{name}: eager {t_e:.3f} ms/step", flush=True)
try:
    gs = tr.capture(x, y, idx, idx.numel())
    l0 = float(gs())
    t_g = wall(tr.replay)
    print(f"loss first replay {l0:.5f} -> after 200+ replays {float(gs.out):.5f}")
    print(f"{name}: hipGraph {t_g:.3f} ms/step, loss {float(gs.out):.5f}", flush=True)
except Exception as ex:  # noqa: BLE001
    print("capture failed:", type(ex).__name__, str(ex)[:300])
