#!/usr/bin/env python3
"""What does hipBLASLt do with the f32 GEMMs of the step when TF32 is allowed (gfx950 has no xf32 MFMA)?  Time and error
vs the plain f32 product and an f64 reference."""
import os
import sys
import time

import torch

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n = 2449029
x = torch.randn(n, 256, generator=g, device=dev)
w = torch.randn(256, 256, generator=g, device=dev) * 0.06
ref = (x[:200000].double() @ w.double().t())


def t(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for mode in ("highest", "high", "medium"):
    torch.set_float32_matmul_precision(mode)
    y = x @ w.t()
    err = float((y[:200000].double() - ref).abs().max() / ref.abs().max())
    print(f"float32_matmul_precision={mode:8s} allow_tf32={torch.backends.cuda.matmul.allow_tf32}: {t(lambda: x @ w.t()):.3f} ms, max rel err vs f64 {err:.2e}", flush=True)
