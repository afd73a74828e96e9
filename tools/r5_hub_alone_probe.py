"""How long does the hub walk take on its own?  K = 256 SpMM-sum of the products-sized graph with the hub launch IN FRONT of the
row walks on the same stream (exact_side_stream = 0: nothing overlaps), one hub launch per aggregate and one per column block —
run under rocprofv3 --kernel-trace --stats for the per-kernel durations; hipEvent time of the whole aggregate printed here."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.layers import calc_gcn_norm
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
n, e, _, _ = DATASETS["products"]
ei = rmat_graph(n, e, seed=0, device=dev)
w = calc_gcn_norm(ei, n).contiguous()
gp = eng.graph_plan(ei, n)
cnt = gp.fwd.counts()
long_mask = cnt > gp.fwd.chunk
print(f"long rows {int(long_mask.sum())}, their edges {int(cnt[long_mask].sum())} of {int(cnt.sum())} ({float(cnt[long_mask].sum()) / float(cnt.sum()):.3f}), longest {int(cnt.max())}, chunk {gp.fwd.chunk}")
x = torch.randn(n, 256, device=dev)
def ev(fn, reps=6):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
with torch.no_grad():
    eng.c_spmm_sum(ei, w, x)
    for side, one, prio in ((0, 1, 0), (0, 0, 0), (1, 1, 0), (1, 0, 0), (1, 1, 1), (1, 0, 1), (1, 1, 0), (1, 0, 0), (1, 1, 1), (1, 0, 1)):
        eng.set_option("exact_side_stream", side); eng.set_option("hub_one_launch", one); eng.set_option("hub_priority", prio)
        print(f"exact_side_stream={side} hub_one_launch={one} hub_priority={prio}: {ev(lambda: eng.c_spmm_sum(ei, w, x), 10):7.3f} ms per aggregate", flush=True)
