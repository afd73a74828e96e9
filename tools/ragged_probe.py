"""ragged rows (K % lane vector != 0) on the products-sized graph: segment sum / max (f32, f16, bf16), gspmm sum / max;
with and without the ragged kernels for the maxima the policy keeps off them (option ragged_max)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev); E = ei.shape[1]; dst = ei[1].contiguous()
w = torch.rand(E, device=dev)
def ev(fn, reps=4):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for dt in (torch.float32, torch.float16, torch.bfloat16):
    for K in (47, 100):
        x = (torch.randn(E, K, device=dev) * 4).to(dt)
        line = f"segment {str(dt)[6:]:9s} K={K:3d}: sum {ev(lambda: eng.c_segment_sum(x, dst, n)):7.3f}"
        for rm in (0, 1):
            eng.set_option("ragged_max", rm)
            line += f"   max(ragged_max={rm}) {ev(lambda: eng.c_segment_max(x, dst, n)):7.3f}"
        eng.set_option("ragged_max", 0)
        print(line, flush=True)
        del x
for K in (41, 47, 100):
    x = torch.randn(n, K, device=dev)
    print(f"gspmm f32 K={K:3d}: sum {ev(lambda: eng.c_spmm_sum(ei, w, x)):7.3f}  max {ev(lambda: eng.c_spmm_max(ei, w, x)):7.3f}", flush=True)
