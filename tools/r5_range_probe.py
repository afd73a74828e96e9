"""Would SOURCE-RANGE phases pay?  One K = 64 column-block walk over 2.45 M destination rows whose edges' sources are confined to
a range of R nodes (R x 256 B = the part of the feature slice the phase gathers from): R = N (2.45 M: 627 MB, what a block
launch gathers from today), N/4 (157 MB: fits the 256 MB Infinity Cache), N/8 (78 MB), N/16 (39 MB), N/64 (10 MB: near the L2s).
Uniform random sources (no hub reuse: the cache effect alone), E = 126 M / (N / R) edges each, so that every case is 'one
phase of a P = N / R phase aggregate'; time x P = what the gathers of a phased aggregate would cost (without the out read-modify-write)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
dev = torch.device("cuda", 0); eng = engine()
N, E_full, K = 2449029, 126167309, 64
def ev(fn, reps=6):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, K, generator=g, device=dev)
for P in (1, 2, 4, 8, 16, 64):
    R, E = N // P, E_full // P
    src = torch.randint(0, R, (E,), generator=g, device=dev)
    dst = torch.randint(0, N, (E,), generator=g, device=dev)
    ei = torch.stack([src, dst]).contiguous()
    w = torch.rand(E, generator=g, device=dev)
    with torch.no_grad():
        eng.c_spmm_sum(ei, w, x)
        t = ev(lambda: eng.c_spmm_sum(ei, w, x))
    print(f"P={P:3d}: sources in [0, {R}) ({R * K * 4 / 1e6:7.1f} MB), E={E}: {t:7.3f} ms per phase, x P = {t * P:7.3f} ms "
          f"({E * (K * 4 + 8) / t / 1e6:7.1f} GB/s algorithmic)", flush=True)
    eng.clear_caches(); del ei, src, dst, w
    torch.cuda.empty_cache()
