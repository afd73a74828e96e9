"""A/B of a register-allocation request (8 wavefronts per SIMD for the f32 sum walks): forward aggregate with the fused
epilogue, plain aggregate, gspmm mean / max forward + backward — products-sized graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev); E = ei.shape[1]
w = torch.rand(E, device=dev)
def ev(fn, reps=6):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
gp = eng.graph_plan(ei, n)
x = torch.randn(n, 256, device=dev)
b = torch.randn(256, device=dev)
with torch.no_grad():
    print(f"{tag} spmm sum K=256            {ev(lambda: eng.c_spmm_sum(ei, w, x)):7.3f}", flush=True)
    print(f"{tag} spmm+bias+relu+drop K=256 {ev(lambda: eng.spmm_epi(gp, w, x, 'sum', bias=b, relu=True, p_drop=0.5)):7.3f}", flush=True)
for K in (64, 256):
    xk = torch.randn(n, K, device=dev, requires_grad=True)
    go = torch.randn(n, K, device=dev)
    for name, fn in (("mean", eng.c_spmm_mean), ("max", eng.c_spmm_max)):
        def fb():
            xk.grad = None
            fn(ei, w, xk).backward(go)
        with torch.no_grad():
            f = ev(lambda: fn(ei, w, xk))
        line = f"{tag} gspmm {name} K={K:3d}: fwd {f:7.3f}  fwd+bwd {ev(fb):7.3f}"
        if name == "max":      # the backward with its witnesses from a compact int32 copy (its own instantiation, MODE_MAXBWD32)
            eng.set_option("maxbwd_arg32", 1)
            line += f"   fwd+bwd (int32 witnesses) {ev(fb):7.3f}"
            eng.set_option("maxbwd_arg32", 0)
        print(line, flush=True)
