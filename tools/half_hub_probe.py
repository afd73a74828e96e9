#!/usr/bin/env python3
"""f16 / bf16 unsorted_segment_sum / mean on the arxiv-sized graph (hub row: ~13 k elements) and on a graph with a
109 k-element hub: the 16-bit sums accumulate in the storage type, so hub rows are walked serially."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()


def timed(fn, reps=9):
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
    return sorted(ts)[len(ts) // 2]


n, e, _, _ = DATASETS["arxiv"]
ei = rmat_graph(n, e, seed=0, device=dev)
dst = ei[1].contiguous()
hub = dst.clone()
hub[:109110] = 5                      # a Reddit-sized hub row on the arxiv-sized edge list
for label, ids in (("arxiv R-MAT (max row %d)" % int(torch.bincount(dst).max()), dst), ("+ a 109 110-element hub", hub)):
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        line = f"{label:32s} {str(dt):15s}"
        for K in (7, 16, 47, 64, 256):
            x = torch.randn(ids.shape[0], K, device=dev).to(dt)
            with torch.no_grad():
                t = timed(lambda: eng.c_segment_sum(x, ids, n))
                line += f"  K={K}: {t:6.3f} ms"
                if dt != torch.float32:     # A/B: the row kernel walking the hubs (round 1) — and the same bits
                    y = eng.c_segment_sum(x, ids, n)
                    eng.hub16 = False
                    t0 = timed(lambda: eng.c_segment_sum(x, ids, n), reps=3)
                    same = torch.equal(y.view(torch.int16), eng.c_segment_sum(x, ids, n).view(torch.int16))
                    eng.hub16 = True
                    line += f" (row walk {t0:6.3f}, {'same bits' if same else 'DIFFERENT'})"
        print(line, flush=True)
