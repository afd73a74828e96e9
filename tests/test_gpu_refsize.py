"""-m gpu: the HIP path against THE REFERENCE ITSELF at benchmark sizes.

`oracle/_ref/_torch_ext.so` is the reference's CPU extension compiled from its own sources (oracle/Makefile); it
travels to the GPU box prebuilt.  Up to round 3 nothing larger than 60 k edges met it — the code paths that only exist
at size (64-column block launches, XCD runs, windowed row order, long-row chunk combine, sorted-weight copies) were
checked HIP-vs-HIP.  Here, on the host cores of the GPU box:

  * arxiv-sized graph (N = 169 343, E = 2.48 M incl. loops), widths 16 / 64 / 256 (the profiler protocol's,
    ops_cpu/ggl_segment_cpu.py:11-12): unsorted_segment_{sum,mean,max} on pre-gathered messages, gspmm sum / mean / max
    forward (+ the sum and max backward walks), bspmm forward + both gradients  — cpu/*.cpp, one core, 0.2-3 s each;
  * products-sized graph (N = 2.45 M, E = 126 M), K = 256: the headline aggregate forward AND its transposed backward
    (spmm_sum_cpu.cpp:29-39, :62-78; ~35 s each);
  * Reddit-sized graph, every 32nd edge: one fused GAT layer (8 x 8) forward + the three gradients against the
    reference ops composed as gat_conv.py:103-112 + softmax.py:29-35 under autograd.

Criterion: oracle/parity.py — rows reduced in one piece bit-identical, chunk-combined hub rows within 1e-5 of the
row's magnitude; argmax bit-exact through the gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; the HIP path has no fallback")
    from gammagl_amd import engine

    return engine()


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ref():
    from oracle import oracle as orc

    r = orc.load_ref_ext()
    if r is None:
        pytest.fail("oracle/_ref/_torch_ext.so is missing: build it with `make -C oracle ref` where /root/reference "
                    "exists (it travels to the GPU box with the snapshot)")
    torch.set_num_threads(1)     # the shipped extension is serial (setup.py:50 never defines its OpenMP macro)
    return r


@pytest.fixture(scope="module")
def arxiv(dev):
    from gammagl_amd.layers import calc_gcn_norm
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["arxiv"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    w = calc_gcn_norm(ei, n).contiguous()
    return ei, w, n


def _one_piece(plan):
    return plan.counts() <= plan.chunk


@pytest.mark.parametrize("K", [16, 64, 256])
def test_arxiv_size_segment_ops_vs_the_reference(eng, dev, ref, arxiv, K):
    """A1-A3 at config 2's size: messages [E, K] -> [N, K]; sum / mean within the criterion, max values AND the
    first-edge-wins argmax (through the gradient the reference's own backward scatters, segment_max.cpp:48-61) exact."""
    from oracle import parity

    ei, _, n = arxiv
    E = int(ei.shape[1])
    g = torch.Generator(device=dev).manual_seed(K)
    msg = torch.randn(E, K, generator=g, device=dev)
    msg[::7] = msg[::7].round()                    # ties for the argmax to break
    go = torch.randn(n, K, generator=g, device=dev)
    ids = ei[1].contiguous()
    plan = eng.seg_plan(ids, n)
    assert plan.n_long > 0, "the arxiv-sized graph has hub rows: the chunk-combine path must be exercised"
    msg_c, ids_c, go_c = msg.cpu(), ids.cpu(), go.cpu()
    one = _one_piece(plan)
    for name in ("sum", "mean"):
        got = getattr(eng, f"c_segment_{name}")(msg, ids, n)
        want = getattr(ref, f"c_segment_{name}")(msg_c, ids_c, n)
        parity.check(got, want, f"segment_{name} K={K}", rows_in_one_piece=one)
    a = msg.clone().requires_grad_(True)
    b = msg_c.clone().requires_grad_(True)
    ya, yb = eng.c_segment_max(a, ids, n), ref.c_segment_max(b, ids_c, n)
    r = parity.report(ya.detach(), yb.detach())
    assert r["rows_bit_exact_frac"] == 1.0, r      # a maximum has no rounding: every row, hub rows included
    ya.backward(go)
    yb.backward(go_c)
    assert torch.equal(a.grad.cpu(), b.grad), "segment_max argmax (witnessed by the gradient) differs from the reference"


@pytest.mark.parametrize("K", [16, 64, 256])
def test_arxiv_size_gspmm_vs_the_reference(eng, dev, ref, arxiv, K):
    """A5-A7 at config 2's size: COO SpMM sum / mean / max forward; the transposed walk of sum and the src-id argmax
    backward of max (spmm_max_cpu.cpp:57-99).  (mean's reference backward calls .item() per (edge, column),
    spmm_mean_cpu.cpp:96 — minutes at this size; its formula is pinned by the small goldens.)"""
    from oracle import parity

    ei, w, n = arxiv
    g = torch.Generator(device=dev).manual_seed(100 + K)
    x = torch.randn(n, K, generator=g, device=dev)
    go = torch.randn(n, K, generator=g, device=dev)
    ei_c, w_c, x_c, go_c = ei.cpu(), w.cpu(), x.cpu(), go.cpu()
    gp = eng.graph_plan(ei, n)
    one, oneT = _one_piece(gp.fwd), _one_piece(gp.bwd)
    for rep in range(2):                           # the second call streams the weights from their sorted copy
        a = x.clone().requires_grad_(True)
        ya = eng.c_spmm_sum(ei, w, a)
        ya.backward(go)
        if rep == 0:
            b = x_c.clone().requires_grad_(True)
            yb = ref.c_spmm_sum(ei_c, w_c, b)
            yb.backward(go_c)
        parity.check(ya.detach(), yb.detach(), f"spmm_sum K={K} call {rep}", rows_in_one_piece=one)
        parity.check(a.grad, b.grad, f"spmm_sum backward K={K} call {rep}", rows_in_one_piece=oneT)
    parity.check(eng.c_spmm_mean(ei, w, x), ref.c_spmm_mean(ei_c, w_c, x_c), f"spmm_mean K={K}", rows_in_one_piece=one)
    a = x.clone().requires_grad_(True)
    b = x_c.clone().requires_grad_(True)
    ya, yb = eng.c_spmm_max(ei, w, a), ref.c_spmm_max(ei_c, w_c, b)
    assert parity.report(ya.detach(), yb.detach())["rows_bit_exact_frac"] == 1.0
    ya.backward(go)
    yb.backward(go_c)
    parity.check(a.grad, b.grad, f"spmm_max backward K={K}", rows_in_one_piece=oneT)


@pytest.mark.parametrize("H,C", [(8, 2), (8, 8), (8, 32)])
def test_arxiv_size_bspmm_vs_the_reference(eng, dev, ref, arxiv, H, C):
    """A8 at config 2's size: multi-head SpMM forward, gx (transposed walk) and gw (one dot per edge and head — no
    reduction over edges, so every element is bit-exact whatever the graph)."""
    from oracle import parity

    ei, _, n = arxiv
    E = int(ei.shape[1])
    g = torch.Generator(device=dev).manual_seed(H * 100 + C)
    x = torch.randn(n, H, C, generator=g, device=dev)
    w = torch.rand(E, H, generator=g, device=dev)
    go = torch.randn(n, H, C, generator=g, device=dev)
    gp = eng.graph_plan(ei, n)
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xb, wb = x.cpu().requires_grad_(True), w.cpu().requires_grad_(True)
    ya = eng.c_bspmm_sum(ei, wa, xa)
    yb = ref.c_bspmm_sum(ei.cpu(), wb, xb)
    ya.backward(go)
    yb.backward(go.cpu())
    parity.check(ya.detach(), yb.detach(), f"bspmm {H}x{C}", rows_in_one_piece=_one_piece(gp.fwd))
    parity.check(xa.grad, xb.grad, f"bspmm gx {H}x{C}", rows_in_one_piece=_one_piece(gp.bwd))
    r = parity.check(wa.grad, wb.grad, f"bspmm gw {H}x{C}")
    # the sorted walk sums a strip's channels in 8-wide slabs when C >= 32 (edgedot.hip): same adds, other association
    if C <= 16:
        assert r["elems_bit_exact_frac"] == 1.0, r


def test_products_size_aggregate_vs_the_reference(eng, dev, ref):
    """The headline kernel at the headline size: ONE K = 256 aggregate of the products-sized graph with GCN norm
    weights, forward (four 64-column block launches over the destination-sorted plan) and the transposed backward,
    against the reference's c_spmm_sum on the same tensors (~35 s per direction on one host core)."""
    if torch.cuda.get_device_properties(dev).total_memory < 100 * 2**30:
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd.layers import calc_gcn_norm
    from gammagl_amd.synth import DATASETS, rmat_graph
    from oracle import parity

    n, e, _, _ = DATASETS["products"]
    K = 256
    ei = rmat_graph(n, e, seed=0, device=dev)
    w = calc_gcn_norm(ei, n).contiguous()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, K, generator=g, device=dev)
    go = torch.randn(n, K, generator=g, device=dev)
    gp = eng.graph_plan(ei, n)
    assert int(eng.lib.ggl_spmm_col_blocks(gp.E, K, n)) == 4 and gp.fwd.n_long > 0
    a = x.clone().requires_grad_(True)
    eng.c_spmm_sum(ei, w, a)                      # first sight of w: gathered in-kernel ...
    ya = eng.c_spmm_sum(ei, w, a)                 # ... second: streamed from its sorted copy (what the training step runs)
    ya.backward(go)
    b = x.cpu().requires_grad_(True)
    yb = ref.c_spmm_sum(ei.cpu(), w.cpu(), b)
    yb.backward(go.cpu())
    rf = parity.check(ya.detach(), yb.detach(), "products K=256 forward", rows_in_one_piece=_one_piece(gp.fwd))
    rb = parity.check(a.grad, b.grad, "products K=256 backward", rows_in_one_piece=_one_piece(gp.bwd))
    # EVERY row is the reference's own bits: the hub rows (longer than the 4096-element chunk) are added up in the
    # reference's serial order by hubf32.hip — a regression in the hub walk (1752 rows here) must fail this test
    assert rf["rows_bit_exact_frac"] == 1.0 and rb["rows_bit_exact_frac"] == 1.0, (rf, rb)
    assert rf["max_abs_err"] == 0.0 and rb["max_abs_err"] == 0.0, (rf, rb)


def test_reddit_size_gat_layer_vs_the_reference_ops(eng, dev, ref):
    """Row G against the reference ops composed the way gat_conv.py:103-112 + softmax.py:29-35 write the layer
    (gather, LeakyReLU, c_segment_max, exp, c_segment_sum, divide, gather * alpha, c_segment_sum), under autograd, on
    every 32nd edge of the Reddit-sized graph (3.6 M edges, its 233 k nodes, hub rows of thousands of edges):
    forward 1e-5, gradients 2e-5 of the row's magnitude — ten times tighter than rounds 2-4 (2e-4), whose looser bound was
    the f32 row sums of g_er, not the exponential: since round 5 the destination walk keeps its sums in double
    (gat_fast.hip gat_bwd_dst2_kernel), and against an fp64 evaluation of the layer every HIP result is CLOSER than the
    reference's own f32 composition (test_gat_gradients_against_an_fp64_ground_truth below)."""
    from gammagl_amd.synth import DATASETS, rmat_graph
    from oracle import parity

    n, e, _, _ = DATASETS["reddit"]
    if torch.cuda.get_device_properties(dev).total_memory < 100 * 2**30:
        e //= 8
    ei = rmat_graph(n, e, seed=0, device=dev)[:, ::32].contiguous()
    H, C = 8, 8
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, H, C, generator=g, device=dev)
    el, er = torch.randn(n, H, generator=g, device=dev), torch.randn(n, H, generator=g, device=dev)
    go = torch.randn(n, H, C, generator=g, device=dev)
    ei, _ = parity.kink_free_edges_logits(ei, el, er)      # (logits within 1e-4 of LeakyReLU's kink left out, oracle/parity.py)
    xa, ela, era = (t.clone().requires_grad_(True) for t in (x, el, er))
    out = eng.gat_fused(ei, ela, era, xa, 0.2)
    out.backward(go)
    xb, elb, erb = (t.cpu().requires_grad_(True) for t in (x, el, er))
    src, dst = ei[0].cpu(), ei[1].cpu()
    s = torch.nn.functional.leaky_relu(elb[src] + erb[dst], 0.2)
    m = ref.c_segment_max(s, dst, n)
    ex = torch.exp(s - m[dst])
    den = ref.c_segment_sum(ex, dst, n)
    alpha = ex / (den[dst] + 1e-16)
    want = ref.c_segment_sum(xb[src] * alpha.unsqueeze(-1), dst, n)
    want.backward(go.cpu())
    parity.check(out.detach(), want.detach(), "fused GAT forward vs the composed reference ops", tol=1e-5)
    parity.check(xa.grad, xb.grad, "fused GAT gx", tol=2e-5)
    # (a logit gradient cancels to exactly 0 over a one-edge row: scale floor = the tensor's mean magnitude)
    parity.check(ela.grad, elb.grad, "fused GAT g_el", tol=2e-5, floor_min=float(elb.grad.abs().mean()))
    parity.check(era.grad, erb.grad, "fused GAT g_er", tol=2e-5, floor_min=float(erb.grad.abs().mean()))


def test_gat_gradients_against_an_fp64_ground_truth(eng, dev, ref):
    """Row G's tolerance, settled with a ground truth (round-4 verdict, weak #2): the fused kernels' gradients agree with the
    reference ops composed in f32 only to ~1e-4 of the row's magnitude — because BOTH are f32 evaluations of a softmax
    gradient (sums of thousands of cancelling terms), not because one of them is wrong.  Measured against the same layer in
    float64: err(HIP) <= max(1e-5, 2 * err(reference f32 composition)) for out, gx, g_el, g_er on the Reddit-sized subgraph
    (every 32nd edge, hub rows of thousands of edges).  The errors are printed: bench.py's config-3 `parity` object
    carries the same figures."""
    from gammagl_amd.synth import DATASETS, rmat_graph
    from oracle import parity

    n, e, _, _ = DATASETS["reddit"]
    if torch.cuda.get_device_properties(dev).total_memory < 100 * 2**30:
        e //= 8
    ei0 = rmat_graph(n, e, seed=0, device=dev)[:, ::32].contiguous()
    for H, C in ((8, 8), (1, 64)):
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn(n, H, C, generator=g, device=dev)
        el, er = torch.randn(n, H, generator=g, device=dev), torch.randn(n, H, generator=g, device=dev)
        go = torch.randn(n, H, C, generator=g, device=dev)
        ei, _ = parity.kink_free_edges_logits(ei0, el, er)     # (logits within 1e-4 of LeakyReLU's kink left out)
        truth = parity.gat_truth_f64(ei, el, er, x, go, n)
        xa, ela, era = (t.clone().requires_grad_(True) for t in (x, el, er))
        out = eng.gat_fused(ei, ela, era, xa, 0.2)
        out.backward(go)
        hip = (out.detach(), xa.grad, ela.grad, era.grad)
        xb, elb, erb = (t.cpu().requires_grad_(True) for t in (x, el, er))
        src, dst = ei[0].cpu(), ei[1].cpu()
        s = torch.nn.functional.leaky_relu(elb[src] + erb[dst], 0.2)
        m = ref.c_segment_max(s, dst, n)
        ex = torch.exp(s - m[dst])
        den = ref.c_segment_sum(ex, dst, n)
        alpha = ex / (den[dst] + 1e-16)
        want = ref.c_segment_sum(xb[src] * alpha.unsqueeze(-1), dst, n)
        want.backward(go.cpu())
        reff = (want.detach(), xb.grad, elb.grad, erb.grad)
        e_hip, e_ref = parity.gat_errors_vs_truth(truth, hip), parity.gat_errors_vs_truth(truth, reff)
        for name in e_hip:
            print(f"GAT {H}x{C} {name}: err vs fp64 truth — HIP {e_hip[name]:.3e}, reference f32 composition {e_ref[name]:.3e}")
            assert e_hip[name] <= max(1e-5, 2.0 * e_ref[name]), (H, C, name, e_hip, e_ref)


def _host_mem_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    return int(line.split()[1]) / 2**20
    except OSError:
        pass
    return 0.0


def _reddit_subgraph(dev, stride):
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["reddit"]
    if torch.cuda.get_device_properties(dev).total_memory < 100 * 2**30:
        e //= 8
    return rmat_graph(n, e, seed=0, device=dev)[:, ::stride].contiguous(), n


def _ref_seg(ref):
    return (lambda s, ids, n: ref.c_segment_max(s, ids, n)), (lambda v, ids, n: ref.c_segment_sum(v, ids, n))


def test_reddit_size_headmean_output_layer_vs_the_reference_ops(eng, dev, ref):
    """Row G, the OUTPUT layer of config 3 — FusedGATConv(64, 41, heads=8, concat=False), the path layers.py sends to the
    ggl_gat_sh_* kernels (aggregate the 64-float input row per head, transform afterwards; 60 % of the Reddit step) — against
    GATConv.forward as gat_conv.py:98-122 writes it (head mean :115-118, bias :120-121) composed from the reference's own
    c_segment_max / c_segment_sum under autograd, on every 32nd edge of the Reddit-sized graph: y, gx, gW, gatt, gbias.
    AND against the same layer in float64: err(HIP) <= max(1e-5, 2 err(reference f32 composition)) for all five.
    (Rounds 3-5 checked this path only against the builder's other kernels at 2e-4 of the tensor's maximum.)"""
    from gammagl_amd.layers import FusedGATConv
    from oracle import parity

    stride = 32 if _host_mem_gb() > 96 else 128      # the composed reference holds ~8 [E, 8, 82] f32 tensors under autograd
    ei, n = _reddit_subgraph(dev, stride)
    F, H, C = 64, 8, 41
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(n, F, generator=g, device=dev)
    W = torch.randn(F, H * C, generator=g, device=dev) * 0.15
    att = torch.randn(1, H, 2 * C, generator=g, device=dev) * 0.2
    bias = torch.randn(C, generator=g, device=dev) * 0.1
    go = torch.randn(n, C, generator=g, device=dev)
    # (edges whose logit lies within 1e-4 of LeakyReLU's kink are left out — oracle/parity.py kink_free_edges: an f32 logit on the
    #  other side of 0 than its float64 value takes the other slope, a jump no precision removes; with them in, HIP and the
    #  reference's composition were BOTH 1.79e-3 from the float64 gx by the same flipped edges)
    ei, dropped = parity.kink_free_edges(ei, x, W, att, H, C)
    print(f"head-mean refsize test: {dropped} near-kink edges of {int(ei.shape[1]) + dropped} left out")
    layer = FusedGATConv(F, C, heads=H, concat=False).to(dev)
    with torch.no_grad():
        layer.w.copy_(W), layer.att.copy_(att), layer.bias.copy_(bias)
    assert eng.gat_headmean_supported(H, F, C), "the head-mean path must be the one under test"
    calls = {"n": 0}
    orig = eng.gat_headmean

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    eng.gat_headmean = counted
    try:
        xa = x.clone().requires_grad_(True)
        y = layer(xa, ei, n)
        y.backward(go)
    finally:
        eng.gat_headmean = orig
    assert calls["n"] == 1, "FusedGATConv(concat=False) did not take the gat_sh_* route"
    hip = (y.detach(), xa.grad, layer.w.grad, layer.att.grad, layer.bias.grad)
    # the reference's f32 composition (host, one core)
    xb, Wb, ab, bb = (t.detach().cpu().requires_grad_(True) for t in (x, W, att, bias))
    yb = parity.gat_conv_composed(xb, Wb, ab, bb, ei.cpu(), n, H, C, concat=False, slope=0.2, seg=_ref_seg(ref))
    yb.backward(go.cpu())
    reff = (yb.detach(), xb.grad, Wb.grad, ab.grad, bb.grad)
    # the float64 truth (GPU, torch scatters)
    xd, Wd, ad, bd = (t.detach().double().requires_grad_(True) for t in (x, W, att, bias))
    yd = parity.gat_conv_composed(xd, Wd, ad, bd, ei, n, H, C, concat=False, slope=0.2)
    yd.backward(go.double())
    truth = (yd.detach(), xd.grad, Wd.grad, ad.grad, bd.grad)
    names = ("y", "gx", "gW", "gatt", "gbias")
    # (gx of a row with ONE in-edge and no out-edge is a pure logit gradient: alpha = 1, d alpha / d e = 0 — it cancels to exactly
    #  0 in the truth, so its scale floor is the tensor's mean magnitude, like g_el / g_er in the hidden-layer tests above)
    e_hip = parity.layer_errors_vs_truth(truth, hip, names, zero_mean_rows=("gx",))
    e_ref = parity.layer_errors_vs_truth(truth, reff, names, zero_mean_rows=("gx",))
    for k in names:
        print(f"head-mean GAT 64 -> 8 x 41 {k}: err vs fp64 truth — HIP {e_hip[k]:.3e}, reference f32 composition {e_ref[k]:.3e}")
    for k in names:
        assert e_hip[k] <= max(1e-5, 2.0 * e_ref[k]), (k, e_hip, e_ref)
    # and f32 against f32, the north_star's figure: 1e-5 of the row's magnitude forward, 2e-5 for the gradients
    parity.check(hip[0], reff[0], "head-mean GAT forward vs the composed reference ops", tol=1e-5)
    parity.check(hip[1], reff[1], "head-mean GAT gx", tol=2e-5, floor_min=float(reff[1].abs().mean()))
    # (the parameter gradients are f32 sums over all 233 k nodes on BOTH sides — each is ~6e-5 of the row's magnitude from the
    #  float64 truth, measured above; against each other they are held to the sum of those two errors, not to a constant)
    for a, b, k in zip(hip[2:], reff[2:], names[2:]):
        a2, b2 = (t.reshape(t.shape[0], -1) if t.dim() > 1 else t.reshape(1, -1) for t in (a, b))
        parity.check(a2, b2.to(dev), f"head-mean GAT {k}", tol=max(2e-5, e_hip[k] + e_ref[k]))


def test_reddit_size_gat_model_vs_the_reference_ops(eng, dev, ref):
    """Config 3's MODEL (models/gat.py:36-72: 8 x 8 concat layer, ELU, head-averaging 41-class output layer; eval mode) forward
    + every parameter gradient on every 64th edge of the Reddit-sized graph: fused HIP layers against the composed reference
    ops and against float64 — the same figures bench.py's config-3 `parity` object carries."""
    from gammagl_amd.layers import GATModel
    from oracle import parity

    stride = 64 if _host_mem_gb() > 96 else 256
    ei, n = _reddit_subgraph(dev, stride)
    torch.manual_seed(5)
    model = GATModel(602, 8, 41, heads=8, drop_rate=0.0, num_layers=2, fused=True).to(dev).eval()
    with torch.no_grad():
        for p in model.parameters():       # trained-like magnitudes (the default init's logits are ~0: a flat softmax)
            p.copy_(torch.randn_like(p) * (0.1 if p.dim() > 1 else 0.05))
    g = torch.Generator(device=dev).manual_seed(12)
    x = torch.randn(n, 602, generator=g, device=dev)
    go = torch.randn(n, 41, generator=g, device=dev)
    params = [(l.w, l.att, l.bias) for l in model.gat_list]
    with torch.no_grad():        # (near-kink edges of either layer left out: oracle/parity.py kink_free_edges)
        ei, dropped = parity.kink_free_edges_model(ei, x, [tuple(p.detach() for p in tpl) for tpl in params], n, 8)
    print(f"GAT model refsize test: {dropped} near-kink edges of {int(ei.shape[1]) + dropped} left out")
    y = model(x, ei, n)
    y.backward(go)
    hip = [y.detach()] + [p.grad for tpl in params for p in tpl]
    names = ["y"] + [f"g{nm}{li}" for li in range(2) for nm in ("W", "att", "b")]

    def run(dtype, device, seg):
        ps = [tuple(p.detach().to(device=device, dtype=dtype).requires_grad_(True) for p in tpl) for tpl in params]
        out = parity.gat_model_composed(x.to(device=device, dtype=dtype), ps, ei.to(device), n, 8, slope=0.2, seg=seg)
        out.backward(go.to(device=device, dtype=dtype))
        return [out.detach()] + [p.grad for tpl in ps for p in tpl]

    reff = run(torch.float32, "cpu", _ref_seg(ref))
    truth = run(torch.float64, dev, None)
    e_hip = parity.layer_errors_vs_truth(truth, hip, names)
    e_ref = parity.layer_errors_vs_truth(truth, reff, names)
    for k in names:
        print(f"GAT model {k}: err vs fp64 truth — HIP {e_hip[k]:.3e}, reference f32 composition {e_ref[k]:.3e}")
        assert e_hip[k] <= max(1e-5, 2.0 * e_ref[k]), (k, e_hip, e_ref)


@pytest.mark.parametrize("K", [16, 64, 256])
def test_arxiv_size_spmm_mean_backward_vs_the_pinned_oracle(eng, dev, oracle, arxiv, K):
    """A6's backward at config 2's size (round-5 verdict, weak #6): gx[src] += g[dst] / count[dst] * w[e] in edge order
    (spmm_mean_cpu.cpp:63-105).  The reference's own loop calls .item() per (edge, column) — minutes at this size — so the
    checker is the pinned C restatement (oracle/ggl_oracle.c spmm_mean_bwd, bit-checked against the reference's output on
    the small goldens): the MODE_MEANBWD walk + prescale at K = 16 / 64 / 256, rows in one piece bit-identical."""
    from oracle import parity

    ei, w, n = arxiv
    g = torch.Generator(device=dev).manual_seed(300 + K)
    x = torch.randn(n, K, generator=g, device=dev)
    go = torch.randn(n, K, generator=g, device=dev)
    gp = eng.graph_plan(ei, n)
    a = x.clone().requires_grad_(True)
    eng.c_spmm_mean(ei, w, a).backward(go)
    ei_n, w_n = ei.cpu().numpy(), w.cpu().numpy()
    _, cnt = oracle.spmm_mean_fwd(ei_n, w_n, x.cpu().numpy())
    want = torch.from_numpy(oracle.spmm_mean_bwd(ei_n, w_n, go.cpu().numpy(), cnt))
    r = parity.check(a.grad, want, f"spmm_mean backward K={K}", rows_in_one_piece=_one_piece(gp.bwd))
    print(f"spmm_mean backward K={K}: {r}")


@pytest.mark.parametrize("K", [16, 64, 256])
def test_arxiv_size_segment_mean_backward_vs_the_reference(eng, dev, ref, oracle, arxiv, K):
    """A2's backward at config 2's size: grad_out[ids] / bincount(ids)[ids] (segment_mean.cpp:44-63) — a gather and one
    divide per element, no reduction: every element bit-exact, against the reference's own autograd AND the C oracle."""
    ei, _, n = arxiv
    E = int(ei.shape[1])
    g = torch.Generator(device=dev).manual_seed(400 + K)
    msg = torch.randn(E, K, generator=g, device=dev)
    go = torch.randn(n, K, generator=g, device=dev)
    ids = ei[1].contiguous()
    a = msg.clone().requires_grad_(True)
    eng.c_segment_mean(a, ids, n).backward(go)
    b = msg.cpu().requires_grad_(True)
    ref.c_segment_mean(b, ids.cpu(), n).backward(go.cpu())
    assert torch.equal(a.grad.cpu(), b.grad), "segment_mean backward differs from the reference's"
    want = torch.from_numpy(oracle.segment_mean_bwd(go.cpu().numpy(), ids.cpu().numpy(), n))
    assert torch.equal(a.grad.cpu(), want), "segment_mean backward differs from the C oracle's"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K", [16, 64])
def test_arxiv_size_16bit_segment_max_vs_the_reference(eng, dev, ref, arxiv, dtype, K):
    """A3 for the 16-bit storage types at config 2's size: values bit-exact (a maximum has no rounding) and the
    first-edge-wins argmax witnessed through the reference's own backward; 16-bit values tie OFTEN (2^16 patterns over
    2.5 M messages), so this is the tie-break's hardest case."""
    ei, _, n = arxiv
    E = int(ei.shape[1])
    g = torch.Generator(device=dev).manual_seed(500 + K)
    msg = torch.randn(E, K, generator=g, device=dev).to(dtype)
    go = torch.randn(n, K, generator=g, device=dev).to(dtype)
    ids = ei[1].contiguous()
    a = msg.clone().requires_grad_(True)
    b = msg.cpu().requires_grad_(True)
    ya, yb = eng.c_segment_max(a, ids, n), ref.c_segment_max(b, ids.cpu(), n)
    assert torch.equal(ya.detach().cpu().view(torch.int16), yb.detach().view(torch.int16))
    ya.backward(go)
    yb.backward(go.cpu())
    assert torch.equal(a.grad.cpu().view(torch.int16), b.grad.view(torch.int16)), "16-bit segment_max argmax differs"
