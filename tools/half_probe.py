#!/usr/bin/env python3
"""f16 / bf16 unsorted_segment_sum on the products-sized graph (a 147 000-element hub row): the ragged 8-wide lanes
for rows that are not 16-byte pieces (K = 47 ...) and the hub rows' serial add chains launched beside the walk over the
other rows (side stream) — A/B against the round-2 paths, f32 beside them.   python tools/half_probe.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.synth import DATASETS, rmat_graph  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
E = ei.shape[1]
dst = ei[1].contiguous()
del ei
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def ev(fn, reps=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


plan = eng.seg_plan(dst, n)
say(f"products-sized R-MAT N={n} E={E}, longest row {plan.max_len}, {plan.n_long} rows above the long-row threshold {plan.chunk}")
for K in (1, 7, 12, 16, 47, 64, 100):
    xf = torch.randn(E, K, device=dev) * 4
    t32 = ev(lambda: eng.c_segment_sum(xf, dst, n))
    line = f"K={K:3d}: f32 {t32:6.2f} ms ({E * (4 * K + 8) / t32 / 1e9:4.2f} TB/s)"
    for dt in (torch.float16, torch.bfloat16):
        x = xf.to(dt)
        res = {}
        for rag, ovl in ((0, False), (1, False), (1, True)):
            eng.set_option("ragged4", rag)
            eng.hub16_overlap = ovl
            res[(rag, ovl)] = ev(lambda: eng.c_segment_sum(x, dst, n))
        eng.set_option("ragged4", 1)
        eng.hub16_overlap = True
        tm = ev(lambda: eng.c_segment_mean(x, dst, n))
        best = res[(1, True)]
        line += (f" | {str(dt)[6:]:8s} round-2 path {res[(0, False)]:6.2f} -> ragged lanes {res[(1, False)]:6.2f} -> + hubs beside "
                 f"{best:6.2f} ms ({E * (2 * K + 8) / best / 1e9:4.2f} TB/s), mean {tm:6.2f}")
        del x
    say(line)
    del xf
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
