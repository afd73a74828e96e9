#!/usr/bin/env python3
"""A few f16 / bf16 segment sums at one width on the products-sized graph, for rocprofv3 --kernel-trace --stats:
which kernel the time goes to (the walk over ordinary rows, the LDS-pipelined hub rows).  python tools/half_one.py K"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 47
n, e, _, _ = DATASETS["products"]
dst = rmat_graph(n, e, seed=0, device=dev)[1].contiguous()
E = dst.shape[0]
for dt in (torch.float16, torch.bfloat16):
    x = (torch.randn(E, K, device=dev) * 4).to(dt)
    for _ in range(3):
        eng.c_segment_sum(x, dst, n)
    torch.cuda.synchronize()
    del x
