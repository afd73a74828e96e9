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
    ei = rmat_graph(n, e, seed=0, device=dev)[:, ::32].contiguous()
    for H, C in ((8, 8), (1, 64)):
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn(n, H, C, generator=g, device=dev)
        el, er = torch.randn(n, H, generator=g, device=dev), torch.randn(n, H, generator=g, device=dev)
        go = torch.randn(n, H, C, generator=g, device=dev)
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
