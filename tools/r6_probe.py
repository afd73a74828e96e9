"""Round-6 probes (one MI355X).  `maxbwd`: gspmm(max) forward + backward through int32 witnesses vs the 1-bit winner mask on the
Reddit-sized graph (dense: 0.48 GB witness matrix at K = 256, cache-resident) and on the products-sized one, K = 128 / 256 / 602 —
the measurement behind ggl_policy_maxbwd_form's footprint gate (round-5 advisor: the mask had only been timed on products)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
what = sys.argv[1:] or ["maxbwd"]


def ev(fn, reps=6):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if "maxbwd" in what:
    for name in ("reddit", "products"):
        n, e, _, _ = DATASETS[name]
        ei = rmat_graph(n, e, seed=0, device=dev)
        E = int(ei.shape[1])
        w = torch.rand(E, device=dev)
        for K in (128, 256, 602):
            xk = torch.randn(n, K, device=dev, requires_grad=True)
            go = torch.randn(n, K, device=dev)

            def fb():
                xk.grad = None
                eng.c_spmm_max(ei, w, xk).backward(go)

            with torch.no_grad():
                f = ev(lambda: eng.c_spmm_max(ei, w, xk))
            form = int(eng.lib.ggl_policy_maxbwd_form(E, n, K))
            line = f"[{name}] gspmm max K={K:3d}: fwd {f:7.3f} ms | policy form {form}"
            for nm, opts in (("int32 witnesses", dict(maxbwd_mask=0)), ("winner mask", dict(maxbwd_mask=1, maxbwd_mask_kmax=0))):
                for k, v in opts.items():
                    eng.set_option(k, v)
                torch.cuda.reset_peak_memory_stats()
                base = torch.cuda.memory_allocated()
                try:
                    t = ev(fb)
                    line += f" | fwd+bwd {nm} {t:7.3f} ms (peak +{(torch.cuda.max_memory_allocated() - base) / 2**30:.1f} GiB)"
                except torch.OutOfMemoryError:
                    line += f" | fwd+bwd {nm} OOM"
                eng.set_option("maxbwd_mask", 128); eng.set_option("maxbwd_mask_kmax", 256)
            print(line, flush=True)
            del xk, go
        eng.clear_caches(); del ei, w
        torch.cuda.empty_cache()

if "gat" in what:
    # the head-mean output layer of config 3 (64 -> 8 x 41) forward + backward and the 2-layer GAT step, round-5 dots +
    # select reduce-scatter (gat_sh_pk = 0) vs packed pair dots + select-free reduce (1), with and without attention dropout
    import torch.nn.functional as Fn
    from gammagl_amd import layers
    n, e, _, _ = DATASETS["reddit"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    x = torch.randn(n, 64, device=dev, requires_grad=True)
    conv = layers.FusedGATConv(64, 41, heads=8, concat=False, dropout_rate=0.6).to(dev); conv.train()
    conv0 = layers.FusedGATConv(64, 41, heads=8, concat=False, dropout_rate=0.0).to(dev); conv0.train()
    xf = torch.randn(n, 602, device=dev); yl = torch.randint(0, 41, (n,), device=dev); tidx = torch.arange(0, n, 3, device=dev)
    for pk, waves in ((0, 0), (1, 0), (1, 4), (0, 0), (1, 0), (1, 4)):
        eng.set_option("gat_sh_pk", pk); eng.set_option("gat_sh_waves", waves)
        f = ev(lambda: conv(x.detach(), ei, n)); fb = ev(lambda: conv(x, ei, n).sum().backward())
        fb0 = ev(lambda: conv0(x, ei, n).sum().backward())
        torch.manual_seed(0)
        net = layers.GATModel(602, 8, 41, 8, 0.6, 2, fused=True).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=5e-4)
        def step():
            net.train(); opt.zero_grad(set_to_none=True)
            Fn.cross_entropy(net(xf, ei, n)[tidx], yl[tidx]).backward(); opt.step()
        print(f"gat_sh_pk={pk} gat_sh_waves={waves}: output layer fwd {f:.2f} ms, fwd+bwd dropout 0.6 {fb:.2f} ms, no dropout {fb0:.2f} ms; 2-layer GAT step {ev(step, 5):.2f} ms", flush=True)
    eng.set_option("gat_sh_pk", 1); eng.set_option("gat_sh_waves", 0)
    eng.clear_caches()

if "planted" in what:
    # a graph WITH locality (hierarchical planted communities, products-sized, native order): K = 256 aggregate under the knobs that
    # decide how much of it reaches the L2s — rows per XCD run (policy: 2048 where locality > 0.5) and the column-block width
    from gammagl_amd.layers import calc_gcn_norm
    from gammagl_amd.synth import DATASETS, planted_pairs
    n = DATASETS["products"][0]
    s, d = planted_pairs(n, seed=0, device=dev)
    loops = torch.arange(n, device=dev)
    ei = torch.stack([torch.cat([s, loops]), torch.cat([d, loops])]).contiguous()
    del s, d
    w = calc_gcn_norm(ei, n).contiguous()
    x = torch.randn(n, 256, device=dev)
    print(f"planted graph: N={n} E={ei.shape[1]}", flush=True)
    with torch.no_grad():
        for run in (-1, 0, 256, 512, 1024, 2048, 4096, 8192, 16384):
            eng.xcd_run_rows = run
            eng.clear_caches()
            gp = eng.graph_plan(ei, n)
            eng.c_spmm_sum(ei, w, x)
            line = f"xcd_run_rows={run:6d} (plan: {int(getattr(gp.fwd, 'xcd_run', 0) or 0):5d}):"
            for cb in (32, 64, 128, 0):
                eng.set_option("col_block", cb)
                line += f"  col_block {cb:3d}: {ev(lambda: eng.c_spmm_sum(ei, w, x)):6.2f} ms"
            eng.set_option("col_block", 64)
            print(line, flush=True)
        eng.xcd_run_rows = -1
    eng.clear_caches()
