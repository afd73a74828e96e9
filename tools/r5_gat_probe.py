"""Round-5 GAT probe (Reddit-sized graph): the output layer (64 -> 8 x 41, heads averaged, attention dropout 0.6) forward and
forward + backward with the source walk built for 3 (default) vs 4 wavefronts per SIMD (option gat_sh_waves), and the whole
2-layer GATModel step of config 3 either way."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine, layers
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
n, e, _, _ = DATASETS["reddit"]
ei = rmat_graph(n, e, seed=0, device=dev)
x = torch.randn(n, 64, device=dev, requires_grad=True)
def ev(fn, reps=7):
    for _ in range(2): fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
conv = layers.FusedGATConv(64, 41, heads=8, concat=False, dropout_rate=0.6).to(dev); conv.train()
conv0 = layers.FusedGATConv(64, 41, heads=8, concat=False, dropout_rate=0.0).to(dev); conv0.train()
xf = torch.randn(n, 602, device=dev); yl = torch.randint(0, 41, (n,), device=dev); tidx = torch.arange(0, n, 3, device=dev)
for waves, zlds, pf, glds in ((0, 1, 1, 0), (0, 1, 1, 0)):
    eng.set_option("gat_sh_waves", waves); eng.set_option("gat_sh_zlds", zlds); eng.set_option("gat_sh_prefetch", pf); eng.set_option("gat_sh_glds", glds)
    f = ev(lambda: conv(x.detach(), ei, n)); fb = ev(lambda: conv(x, ei, n).sum().backward())
    fb0 = ev(lambda: conv0(x, ei, n).sum().backward())
    torch.manual_seed(0)
    net = layers.GATModel(602, 8, 41, 8, 0.6, 2, fused=True).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=5e-4)
    def step():
        net.train(); opt.zero_grad(set_to_none=True)
        F.cross_entropy(net(xf, ei, n)[tidx], yl[tidx]).backward(); opt.step()
    print(f"gat_sh_waves={waves} gat_sh_zlds={zlds} gat_sh_prefetch={pf} gat_sh_glds={glds}: output layer fwd {f:.2f} ms, fwd+bwd {fb:.2f} ms (no dropout: {fb0:.2f}); 2-layer GAT step {ev(step, 5):.2f} ms", flush=True)
eng.set_option("gat_sh_waves", 0); eng.set_option("gat_sh_zlds", 0); eng.set_option("gat_sh_prefetch", 1); eng.set_option("gat_sh_zlds", 1); eng.set_option("gat_sh_glds", 0)
