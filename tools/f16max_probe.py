"""f16 / bf16 segment sum and max at K = 32 / 64 / 128 on the products-sized graph (the 8-per-lane walks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev); E = ei.shape[1]; dst = ei[1].contiguous()
def ev(fn, reps=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for dt in (torch.float16, torch.bfloat16):
    for K in (32, 64, 128):
        x = (torch.randn(E, K, device=dev) * 4).to(dt)
        print(f"{str(dt)[6:]:9s} K={K:3d}: sum {ev(lambda: eng.c_segment_sum(x, dst, n)):7.3f}  max {ev(lambda: eng.c_segment_max(x, dst, n)):7.3f}", flush=True)
        del x
