"""A/B of the 16-deep unroll for narrow-row max walks (option unroll_narrow_max) on the products-sized graph."""
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
for dt in (torch.float32, torch.float16, torch.bfloat16, torch.int32):
    for K in (1, 4, 8, 16):
        x = (torch.randn(E, K, device=dev) * 4).to(dt)
        res = []
        for knob in (0, 1):
            eng.set_option("unroll_narrow_max", knob)
            res.append(ev(lambda: eng.c_segment_max(x, dst, n)))
        eng.set_option("unroll_narrow_max", 0)
        s = ev(lambda: eng.c_segment_sum(x, dst, n))
        print(f"{str(dt)[6:]:9s} K={K:2d}: max U=4 {res[0]:7.3f} ms   max U=16 {res[1]:7.3f} ms   (sum {s:7.3f} ms)", flush=True)
        del x
