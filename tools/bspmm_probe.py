"""bspmm forward / forward+backward on the products-sized graph (gradients dropped between iterations) + two K = 256 gspmm
sums; run once per setting of GGL_EXACT_SIDE_PRIORITY."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev); E = ei.shape[1]
def ev(fn, reps=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
tag = "prio=" + os.environ.get("GGL_EXACT_SIDE_PRIORITY", "0")
for H, C in ((8, 8), (16, 16), (32, 8), (1, 256), (8, 32)):
    x = torch.randn(n, H, C, device=dev, requires_grad=True)
    wh = torch.rand(E, H, device=dev, requires_grad=True)
    go = torch.randn(n, H, C, device=dev)
    f = ev(lambda: eng.c_bspmm_sum(ei, wh.detach(), x.detach()))
    def fb():
        x.grad = None; wh.grad = None
        eng.c_bspmm_sum(ei, wh, x).backward(go)
    line = f"{tag} bspmm H={H:2d} C={C:3d}: fwd {f:7.3f}"
    for sw in (True, False):
        eng.gradw_sorted = sw
        t = ev(fb)
        line += f"   fwd+bwd ({'policy' if sw else 'COO gw'}) {t:7.3f} ({t / f:4.2f}x)"
    eng.gradw_sorted = True
    print(line, flush=True)
    del x, wh, go
w = torch.rand(E, device=dev)
x = torch.randn(n, 256, device=dev)
with torch.no_grad():
    eng.set_option("col_block", 0)
    print(f"{tag} gspmm sum K=256 one launch  {ev(lambda: eng.c_spmm_sum(ei, w, x)):7.3f}", flush=True)
    eng.set_option("col_block", 64)
    print(f"{tag} gspmm sum K=256 4 x 64 cols {ev(lambda: eng.c_spmm_sum(ei, w, x)):7.3f}", flush=True)
