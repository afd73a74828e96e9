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

if "gatfull" in what:
    # full Reddit-sized graph, output layer 64 -> 8 x 41 (default init, no dropout): head-mean path with gat_sh_pk = 1 / 0 and the
    # transform-first kernels, pairwise — which differences are the packed form's and which are two f32 evaluations' own
    from gammagl_amd import layers
    from oracle import parity
    n, e, _, _ = DATASETS["reddit"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(n, 64, generator=g, device=dev)
    go = torch.randn(n, 41, generator=g, device=dev)
    torch.manual_seed(0)
    fg = layers.FusedGATConv(64, 41, heads=8, concat=False).to(dev)
    res = {}
    for name, fast, pk in (("headmean pk=1", True, 1), ("headmean pk=0", True, 0), ("transform-first", False, 1)):
        eng.gat_fast = fast; eng.set_option("gat_sh_pk", pk)
        for p_ in fg.parameters(): p_.grad = None
        xa = x.clone().requires_grad_(True)
        y = fg(xa, ei, n); y.backward(go)
        res[name] = [y.detach(), xa.grad, fg.w.grad.clone(), fg.att.grad.clone()]
    eng.gat_fast = True; eng.set_option("gat_sh_pk", 1)
    names = list(res)
    for i in range(3):
        for j in range(i + 1, 3):
            line = f"{names[i]} vs {names[j]}:"
            for a, b, nm in zip(res[names[i]], res[names[j]], ("y", "gx", "gW", "gatt")):
                a2, b2 = (t.reshape(t.shape[0], -1) if t.dim() > 1 else t.reshape(1, -1) for t in (a, b))
                r = parity.report(a2, b2, tol=1.0, floor_min=float(b2.abs().mean()))
                line += f"  {nm} row-scale {r['max_rel_err']:.2e} (abs {r['max_abs_err']:.2e}, max|b| {float(b2.abs().max()):.2e}, mean|b| {float(b2.abs().mean()):.2e})"
            print(line, flush=True)
    eng.clear_caches()

if "gatdbg" in what:
    from gammagl_amd import layers
    n, e, _, _ = DATASETS["reddit"]
    for stride in (64, 8, 1):
        ei = rmat_graph(n, e, seed=0, device=dev)[:, ::stride].contiguous()
        gp = eng.graph_plan(ei, n)
        print(f"stride {stride}: E={ei.shape[1]} fwd n_long={gp.fwd.n_long} n_chunks={gp.fwd.n_chunks} chunk={gp.fwd.chunk} | bwd n_long={gp.bwd.n_long} n_chunks={gp.bwd.n_chunks}", flush=True)
        x = torch.randn(n, 64, device=dev, requires_grad=True)
        fg = layers.FusedGATConv(64, 41, heads=8, concat=False).to(dev)
        y = fg(x, ei, n); torch.cuda.synchronize(); print("  fwd ok", flush=True)
        y.sum().backward(); torch.cuda.synchronize(); print("  bwd ok", float(x.grad.abs().max()), flush=True)
        eng.clear_caches()

if "gatacc" in what:
    # WHERE the head-mean layer's gradient error at 14 M edges comes from: its internal T / gel / ger (and A, den of the forward)
    # against float64 evaluations of the same quantities
    import ctypes
    from gammagl_amd.ops import _ptr
    n, e, _, _ = DATASETS["reddit"]
    stride = int(os.environ.get("STRIDE", "8"))
    ei = rmat_graph(n, e, seed=0, device=dev)[:, ::stride].contiguous()
    F, H, C = 64, 8, 41
    g = torch.Generator(device=dev).manual_seed(21)
    x = torch.randn(n, F, generator=g, device=dev)
    W = torch.randn(F, H * C, generator=g, device=dev) * 0.15
    att = torch.randn(1, H, 2 * C, generator=g, device=dev) * 0.2
    gy = torch.randn(n, C, generator=g, device=dev)
    src, dst = ei[0], ei[1]
    # ---- float64 truths of the intermediates
    xd, Wd, ad = x.double(), W.double(), att.double()
    zd = (xd @ Wd).view(n, H, C)
    eld, erd = (zd * ad[:, :, :C]).sum(-1), (zd * ad[:, :, C:]).sum(-1)
    raw = eld[src] + erd[dst]
    ed = torch.nn.functional.leaky_relu(raw, 0.2)
    md = torch.full((n, H), -float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, dst.view(-1, 1).expand_as(ed), ed, reduce="amax")
    exd = torch.exp(ed - md[dst]); dend = torch.zeros(n, H, dtype=torch.float64, device=dev).index_add_(0, dst, exd)
    alpha = exd / (dend[dst] + 1e-16)                                    # [E, H]
    gyh = gy.double() / H
    T_true = torch.zeros(n, H, C, dtype=torch.float64, device=dev)
    da = torch.empty_like(alpha)
    for h in range(H):
        T_true[:, h, :].index_add_(0, src, alpha[:, h:h + 1] * gyh[dst])
        da[:, h] = (gyh[dst] * zd[src, h, :]).sum(-1)                     # <gy_i / H, z_jh>
    s = torch.zeros(n, H, dtype=torch.float64, device=dev).index_add_(0, dst, alpha * da)
    de = alpha * (da - s[dst]) * torch.where(raw > 0, 1.0, 0.2)
    gel_true = torch.zeros(n, H, dtype=torch.float64, device=dev).index_add_(0, src, de)
    ger_true = torch.zeros(n, H, dtype=torch.float64, device=dev).index_add_(0, dst, de)
    del exd, raw, ed
    # ---- the library's pieces (GATHeadMean.forward / backward, ops.py)
    gp = eng.graph_plan(ei, n)
    Wr = W.view(F, H, C); a_src, a_dst = att[0, :, :C], att[0, :, C:]
    U, V = (Wr * a_src).sum(-1), (Wr * a_dst).sum(-1)
    el, er = (x @ U).contiguous(), (x @ V).contiguous()
    rowmax = torch.empty((n, H), device=dev); den = torch.empty((n, H), device=dev); A = torch.empty((n, H, F), device=dev)
    part = torch.empty(eng.lib.ggl_gat_sh_partial_bytes(gp.fwd.n_chunks, F) + 16, dtype=torch.uint8, device=dev) if gp.fwd.n_long else None
    cs = gp.fwd.c_struct(part)
    eng._check(eng.lib.ggl_gat_sh_fwd(ctypes.byref(cs), _ptr(gp.col), _ptr(el), _ptr(er), _ptr(x), F, 0.2, 0.0, None, _ptr(rowmax), _ptr(A), _ptr(den), eng._stream(dev)))
    Cp = 44
    Wst = Wr.permute(1, 0, 2).reshape(H * F, C)
    gyh32 = gy / H
    gyp = torch.nn.functional.pad(gyh32, (0, Cp - C)).contiguous()
    G = (gyh32 @ Wst.t()).view(n, H, F).contiguous()
    stats = torch.empty((n, H, 4), device=dev)
    eng._check(eng.lib.ggl_gat_sh_stats(_ptr(er), _ptr(rowmax), _ptr(den), _ptr(G), _ptr(A), n, F, _ptr(stats), eng._stream(dev)))
    z = torch.nn.functional.pad((x @ W).view(n, H, C), (0, Cp - C)).contiguous()
    ger = torch.empty((n, H), device=dev); gel = torch.empty((n, H), device=dev); T = torch.empty((n, H, Cp), device=dev)
    bwd = gp.bwd
    part_f = torch.empty(eng.lib.ggl_gat_sh_partial_bytes(gp.fwd.n_chunks, 8) + 16, dtype=torch.uint8, device=dev) if gp.fwd.n_long else None
    part_t = torch.empty(eng.lib.ggl_gat_sh_partial_bytes(bwd.n_chunks, Cp) + 16, dtype=torch.uint8, device=dev) if bwd.n_long else None
    cs, csT = gp.fwd.c_struct(part_f), bwd.c_struct(part_t)
    eng._check(eng.lib.ggl_gat_sh_bwd(ctypes.byref(cs), _ptr(gp.col), ctypes.byref(csT), _ptr(gp.colT), None, _ptr(el), _ptr(x), F, _ptr(G), _ptr(stats), _ptr(z),
                                      _ptr(gyp), Cp, 0.2, 0.0, None, _ptr(ger), _ptr(T), _ptr(gel), eng._stream(dev)))
    torch.cuda.synchronize()
    from oracle import parity
    def err(a, t, nm):
        a2, t2 = a.double().reshape(a.shape[0], -1), t.reshape(t.shape[0], -1)
        r = parity.report(a2, t2, tol=1.0, floor_min=float(t2.abs().mean()))
        print(f"  {nm:28s} row-scale err {r['max_rel_err']:.3e}  abs {r['max_abs_err']:.3e}  mean|truth| {float(t2.abs().mean()):.3e}", flush=True)
    print(f"stride {stride}: E={ei.shape[1]}, longest row {int(gp.fwd.counts().max())}", flush=True)
    err(el, eld, "el = x (W a_src)"); err(er, erd, "er")
    err(rowmax, md, "rowmax"); err(den, dend, "den")
    s32 = stats[:, :, 3]
    err(s32, s, "s_i = <G_i, A_i> (stats.w)")
    err(T[:, :, :C], T_true, "T (source walk)"); err(gel, gel_true, "gel (source walk)"); err(ger, ger_true, "ger (destination walk)")
    # the same logit gradients by plain f32 torch ops (the yardstick)
    raw32 = el[src] + er[dst]; e32 = torch.nn.functional.leaky_relu(raw32, 0.2)
    m32 = torch.full((n, H), -float("inf"), device=dev).scatter_reduce(0, dst.view(-1, 1).expand_as(e32), e32, reduce="amax")
    ex32 = torch.exp(e32 - m32[dst]); den32 = torch.zeros(n, H, device=dev).index_add_(0, dst, ex32)
    al32 = ex32 / (den32[dst] + 1e-16)
    da32 = torch.empty_like(al32)
    for h in range(H):
        da32[:, h] = (gyh32[dst] * z[src, h, :C]).sum(-1)
    s_32 = torch.zeros(n, H, device=dev).index_add_(0, dst, al32 * da32)
    de32 = al32 * (da32 - s_32[dst]) * torch.where(raw32 > 0, 1.0, 0.2)
    err(torch.zeros(n, H, device=dev).index_add_(0, src, de32), gel_true, "gel, plain torch f32")
    err(torch.zeros(n, H, device=dev).index_add_(0, dst, de32), ger_true, "ger, plain torch f32")
    err(s_32, s, "s_i, plain torch f32")
    eng.clear_caches()

if "arxivchunk" in what:
    # arxiv-sized graph: the K = 256 aggregate under different long-row thresholds (hub walk vs row walk balance)
    from gammagl_amd.layers import calc_gcn_norm
    n, e, _, _ = DATASETS["arxiv"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    w = calc_gcn_norm(ei, n).contiguous()
    old = eng.chunk
    with torch.no_grad():
        for K in (256, 128, 48):
            x = torch.randn(n, K, device=dev)
            line = f"arxiv K={K}:"
            for chunk in (0, 128, 256, 512, 1024, 2048, 4096):
                eng.chunk = chunk; eng.clear_caches()
                gp = eng.graph_plan(ei, n)
                eng.c_spmm_sum(ei, w, x); eng.c_spmm_sum(ei, w, x)
                line += f"  chunk {chunk if chunk else 'auto(' + str(gp.fwd.chunk) + ')'}: {ev(lambda: eng.c_spmm_sum(ei, w, x), 20) * 1e3:6.1f} us (n_long {gp.fwd.n_long})"
            print(line, flush=True)
    eng.chunk = old; eng.clear_caches()

if "prodchunk" in what:
    # products-sized graph: the K = 256 aggregate under different long-row thresholds (which rows the serial hub walk takes)
    from gammagl_amd.layers import calc_gcn_norm
    n, e, _, _ = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    w = calc_gcn_norm(ei, n).contiguous()
    x = torch.randn(n, 256, device=dev)
    old = eng.chunk
    with torch.no_grad():
        for chunk in (0, 512, 1024, 2048, 4096, 8192, 16384):
            eng.chunk = chunk; eng.clear_caches()
            gp = eng.graph_plan(ei, n)
            for _ in range(3): eng.c_spmm_sum(ei, w, x)
            t = ev(lambda: eng.c_spmm_sum(ei, w, x), 10)
            cnt = gp.fwd.counts()
            share = float(cnt[cnt > gp.fwd.chunk].sum()) / float(cnt.sum())
            print(f"products K=256 chunk {chunk if chunk else 'auto(' + str(gp.fwd.chunk) + ')'}: {t:7.3f} ms  (n_long {gp.fwd.n_long}, {share * 100:.1f} % of the edges in long rows)", flush=True)
    eng.chunk = old; eng.clear_caches()

if "phases" in what:
    # SOURCE-RANGE PHASES on the real (skewed) products-sized graph: the K = 256 aggregate as P accumulating launches, phase p over the
    # edges whose source lies in the p-th slice of the id range (edges arrive sorted by source, so a row's phases continue its serial
    # chain in the reference's order: accumulate mode seeds the row from `out`).  Round 5's range probe used UNIFORM sources; here
    # every slice keeps its share of the hubs, so an XCD's L2 holds the 16 K hottest rows of 1/P of the sources instead of all of them.
    from gammagl_amd.layers import calc_gcn_norm
    from oracle import parity
    n, e, _, _ = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    w = calc_gcn_norm(ei, n).contiguous()
    E0 = int(ei.shape[1]) - n                       # rmat_graph: E0 edges sorted by source, then the n self-loops (add_self_loops order)
    src = ei[0, :E0].contiguous()
    assert bool((src[1:] >= src[:-1]).all()), "edge list is not sorted by source"
    assert bool((ei[0, E0:] == ei[1, E0:]).all())
    x = torch.randn(n, 256, device=dev)
    with torch.no_grad():
        ref = eng.c_spmm_sum(ei, w, x); ref = eng.c_spmm_sum(ei, w, x)
        t1 = ev(lambda: eng.c_spmm_sum(ei, w, x), 8)
        print(f"single plan: {t1:7.3f} ms", flush=True)
        for P in (2, 3, 4, 6, 8):
            plans = []
            for p in range(P):
                lo, hi = (n * p) // P, (n * (p + 1)) // P
                a = int(torch.searchsorted(src, torch.tensor([lo], device=dev))[0]); b = int(torch.searchsorted(src, torch.tensor([hi], device=dev))[0])
                ei_p = ei[:, a:b]; w_p = w[a:b]
                if p == P - 1:                           # the loops come last in every row's original order: behind the last slice
                    ei_p = torch.cat([ei_p, ei[:, E0:]], dim=1); w_p = torch.cat([w_p, w[E0:]])
                ei_p = ei_p.contiguous(); w_p = w_p.contiguous()
                plans.append((eng.graph_plan(ei_p, n, n), w_p, ei_p))
            out = torch.empty_like(x)
            def run():
                for p, (gp, w_p, _) in enumerate(plans):
                    eng.spmm_sum_into(gp.fwd, gp.col, w_p, x, out, accumulate=(p > 0))
            run(); run(); run()
            t = ev(run, 8)
            r = parity.report(out, ref)
            nl = [int(gp.fwd.n_long) for gp, _, _ in plans]
            print(f"P={P}: {t:7.3f} ms  rows bit-identical to the single walk {r['rows_bit_exact_frac']:.6f}  max row-scale err {r['max_rel_err']:.2e}  long rows per phase {nl}", flush=True)
            del plans, out
            eng.clear_caches(); torch.cuda.empty_cache()
