"""-m gpu: every single-GPU BASELINE config at its FULL size, through size-independent properties (the CPU
oracle would take minutes there): config 3 (Reddit-sized fused GAT, hub-row chunk path, both head shapes of
the model), config 4 (products-sized neighbour sampling [25, 10] + SAGEConv(mean) blocks), config 5 (one rank's
share of an 8-way partition built without a global edge list, halo bookkeeping checked against the global SpMM)."""
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


def _big(dev):
    return torch.cuda.get_device_properties(dev).total_memory >= 100 * 2**30


@pytest.fixture(scope="module")
def reddit(dev):
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["reddit"]
    return rmat_graph(n, e, seed=0, device=dev), n


def _gat_rows_f64(ei_plan, rows, el, er, x, slope):
    """out rows of the in-tree GATConv math (gat_conv.py:103-112 + softmax.py:29-35) in f64 for a row sample."""
    gp = ei_plan
    rp = gp.fwd.rowptr
    outs = []
    for r in rows.tolist():
        b, e = int(rp[r]), int(rp[r + 1])
        src = gp.col[b:e].long()
        s = torch.nn.functional.leaky_relu(el[src].double() + er[r].double(), slope)       # [deg, H]
        a = torch.softmax(s, 0) if e > b else s
        outs.append((a.unsqueeze(-1) * x[src].double()).sum(0))
    return torch.stack(outs)


@pytest.mark.parametrize("H,C", [(8, 8), (8, 41)])
def test_reddit_size_fused_gat(eng, dev, reddit, H, C):
    """Config 3 at full size (N = 232 965, E = 114.8 M + loops): hub rows take the chunked path; the op is linear
    in x for fixed logits (adjoint identity in f64), softmax weights sum to 1 (x = 1 -> out = 1, zero logit
    gradients), a row sample incl. the heaviest row equals the in-tree math in f64, and (8 x 8) the fast kernels
    agree with the generic ones, forward and all three gradients."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    ei, N = reddit
    gp = eng.graph_plan(ei, N)
    assert gp.fwd.n_long > 0 and gp.fwd.max_len > 50_000       # the hub rows are reduced chunk-wise
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, generator=g, device=dev)     # noqa: E731
    x, el, er, go = mk(N, H, C), mk(N, H), mk(N, H), mk(N, H, C)
    xa, ela, era = (t.clone().requires_grad_(True) for t in (x, el, er))
    out = eng.gat_fused(ei, ela, era, xa, 0.2)
    out.backward(go)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(xa.grad).all())
    lhs = (out.detach().double() * go.double()).sum()
    rhs = (x.double() * xa.grad.double()).sum()
    torch.testing.assert_close(lhs, rhs, rtol=1e-5, atol=1e-2)
    # row sample against the in-tree math in f64 (heaviest row + 48 random rows)
    cnt = gp.fwd.counts()
    rows = torch.cat([cnt.argmax().reshape(1), torch.randint(0, N, (48,), generator=g, device=dev)])
    ref = _gat_rows_f64(gp, rows, el, er, x, 0.2)
    bound = 1e-5 * x.abs().max().double() + 1e-6
    assert float((out.detach()[rows].double() - ref).abs().max()) <= float(bound) * 4
    # partition of unity
    ones = torch.ones(N, H, C, device=dev)
    e1, e2 = el.clone().requires_grad_(True), er.clone().requires_grad_(True)
    o1 = eng.gat_fused(ei, e1, e2, ones, 0.2)
    assert float((o1.detach() - 1).abs().max()) < 1e-5
    o1.backward(go)
    scale = float(go.abs().mean()) * float(cnt.float().mean())
    assert float(e1.grad.abs().max()) < 1e-3 * scale and float(e2.grad.abs().max()) < 1e-3 * scale
    if eng.lib.ggl_gat_fast_supported(H, C):
        eng.gat_fast = False
        try:
            xb, elb, erb = (t.clone().requires_grad_(True) for t in (x, el, er))
            outb = eng.gat_fused(ei, elb, erb, xb, 0.2)
            outb.backward(go)
        finally:
            eng.gat_fast = True
        torch.testing.assert_close(out.detach(), outb.detach(), rtol=1e-5, atol=1e-5)
        for a, b in ((xa.grad, xb.grad), (ela.grad, elb.grad), (era.grad, erb.grad)):
            tol = 1e-4 * float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= tol, (float((a - b).abs().max()), tol)
    else:
        assert C == 41


def test_reddit_size_gat_output_layer_headmean(eng, dev, reddit):
    """Config 3's OUTPUT layer at full size (64 hidden -> 8 heads x 41 classes, heads averaged): the
    aggregate-then-transform path == the transform-then-aggregate kernels, output and every gradient."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd import layers

    ei, N = reddit
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(N, 64, generator=g, device=dev)
    go = torch.randn(N, 41, generator=g, device=dev)
    torch.manual_seed(0)
    fg = layers.FusedGATConv(64, 41, heads=8, concat=False).to(dev)
    assert eng.gat_headmean_supported(8, 64, 41)
    res = []
    try:
        for fast in (True, False):
            eng.gat_fast = fast
            for p_ in fg.parameters():
                p_.grad = None
            xa = x.clone().requires_grad_(True)
            y = fg(xa, ei, N)
            y.backward(go)
            res.append([y.detach(), xa.grad, fg.w.grad.clone(), fg.att.grad.clone()])
    finally:
        eng.gat_fast = True
    # The partner here — 8 x 44 padded channels = 352 columns — is NOT on the fast kernels (H C <= 256): it is the round-1 generic
    # pair with an [E, H, 2] alpha / de buffer and f32 row sums, whose own gradients sit ~1e-4 of the tensor's maximum from an exact
    # evaluation on 10^5-edge rows.  So: the OUTPUT at the row-scale criterion (2e-5; measured 2.2e-6), the gradients at 3e-4 of the
    # tensor's maximum as in rounds 3-5 (measured: gx 1.2e-4, gW 4.3e-5, gatt 2.3e-4 of the maximum; round 6 probed both packed /
    # unpacked forms of the head-mean walks and its double row sums against this partner: the differences did not move by a bit,
    # i.e. they are the partner's — profiles/r6_gat_fullsize_forms.txt).  The head-mean path's CORRECTNESS is pinned against the
    # reference ops and float64 in test_gpu_refsize.py (3.6 M edges) and below at 14 M edges (rows of 19 k edges).
    from oracle import parity

    r = parity.report(res[0][0], res[1][0], tol=2e-5)
    print(f"full-size head-mean vs transform-first y: row-scale {r['max_rel_err']:.3e}")
    assert r["ok"], r
    for a, b, nm in zip(res[0][1:], res[1][1:], ("gx", "gW", "gatt")):
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        print(f"full-size head-mean vs transform-first {nm}: {err / scale:.3e} of the tensor's maximum")
        assert err <= 3e-4 * scale + 1e-6, (nm, err, scale)
    assert bool(torch.isfinite(res[0][0]).all())


def test_reddit_eighth_headmean_layer_vs_float64(eng, dev):
    """The head-mean output layer (ggl_gat_sh_*) against GROUND TRUTH at the largest size a float64 evaluation fits in 288 GB:
    every 8th edge of the Reddit-sized graph (14.4 M edges, hub rows of 19 k edges, hub chunks of 1024) — the layer in float64
    (oracle/parity.py gat_conv_lean: torch scatters) and, as the yardstick, the same composition in float32 (torch ops, none of this
    library's kernels): err(HIP) <= max(1e-5, 2 err(torch f32)) for y, gx, gW, gatt, gbias."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd import layers
    from gammagl_amd.synth import DATASETS, rmat_graph
    from oracle import parity

    n, e, _, _ = DATASETS["reddit"]
    ei = rmat_graph(n, e, seed=0, device=dev)[:, ::8].contiguous()
    F, H, C = 64, 8, 41
    g = torch.Generator(device=dev).manual_seed(21)
    x = torch.randn(n, F, generator=g, device=dev)
    W = torch.randn(F, H * C, generator=g, device=dev) * 0.15
    att = torch.randn(1, H, 2 * C, generator=g, device=dev) * 0.2
    bias = torch.randn(C, generator=g, device=dev) * 0.1
    go = torch.randn(n, C, generator=g, device=dev)
    # (edges whose logit is within 1e-4 of LeakyReLU's kink are left out: oracle/parity.py kink_free_edges — at this size a few
    #  dozen f32 logits fall on the other side of 0 than their float64 values, and BOTH f32 evaluations then sit 7e-2 from the
    #  "truth" by the same jump)
    ei, dropped = parity.kink_free_edges(ei, x, W, att, H, C)
    print(f"14 M-edge test: {dropped} near-kink edges of {int(ei.shape[1]) + dropped} left out")
    layer = layers.FusedGATConv(F, C, heads=H, concat=False).to(dev)
    with torch.no_grad():
        layer.w.copy_(W), layer.att.copy_(att), layer.bias.copy_(bias)
    assert eng.gat_headmean_supported(H, F, C)
    xa = x.clone().requires_grad_(True)
    y = layer(xa, ei, n)
    y.backward(go)
    hip = (y.detach(), xa.grad, layer.w.grad, layer.att.grad, layer.bias.grad)
    eng.clear_caches()

    def composed(dtype):
        ps = [t.detach().to(dtype).requires_grad_(True) for t in (x, W, att, bias)]
        out = parity.gat_conv_lean(*ps, ei, n, H, C, concat=False, slope=0.2)
        out.backward(go.to(dtype))
        res_ = [out.detach()] + [p.grad for p in ps]
        del out, ps
        torch.cuda.empty_cache()
        return res_

    f32 = composed(torch.float32)
    truth = composed(torch.float64)
    names = ("y", "gx", "gW", "gatt", "gbias")
    e_hip = parity.layer_errors_vs_truth(truth, hip, names, zero_mean_rows=("gx",))
    e_f32 = parity.layer_errors_vs_truth(truth, f32, names, zero_mean_rows=("gx",))
    for k in names:
        print(f"head-mean GAT at 14 M edges {k}: err vs fp64 truth — HIP {e_hip[k]:.3e}, torch f32 composition {e_f32[k]:.3e}")
    for k in names:
        assert e_hip[k] <= max(1e-5, 2.0 * e_f32[k]), (k, e_hip, e_f32)


def test_products_size_sampler_and_sage_blocks(eng, dev):
    """Config 4 at full size: NeighborSampler([25, 10]) over the products-sized CSR, 2048 seeds — block structure
    invariants, every block edge is a real edge between the right nodes, the block aggregate straight from the
    sampler's CSR == unsorted_segment_mean on a freshly built plan (bit for bit), SAGEConv(mean) with its fused
    epilogue == the written-out formula, and consecutive batches draw independently."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd import layers
    from gammagl_amd.sampler import NeighborSampler
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    ns = NeighborSampler(ei, [25, 10], num_nodes=n, eng=eng)
    g = torch.Generator(device=dev).manual_seed(3)
    seeds = torch.randperm(n, generator=g, device=dev)[:2048]
    batch, n_id, adjs = ns.sample(seeds)
    assert torch.equal(batch, seeds) and torch.equal(n_id[:2048], seeds) and n_id.unique().numel() == n_id.numel()
    assert adjs[1].size[1] == 2048 and adjs[0].size[1] == adjs[1].size[0] and adjs[0].size[0] == n_id.numel()
    deg = ns.rowptr[1:] - ns.rowptr[:-1]
    for adj, fan in zip(adjs, (10, 25)):
        src_l, dst_l = adj.edge_index
        n_src, n_dst = adj.size
        assert int(src_l.max()) < n_src and int(dst_l.max()) < n_dst and bool((dst_l[1:] >= dst_l[:-1]).all())
        per_row = adj.rowptr[1:] - adj.rowptr[:-1]
        assert torch.equal(per_row, deg[n_id[:n_dst]].clamp(max=fan))
        # a real edge of the graph between the right global nodes, no edge twice in a row
        assert torch.equal(ei[0][adj.e_id], n_id[src_l]) and torch.equal(ei[1][adj.e_id], n_id[dst_l])
        key = dst_l * n_src + src_l
        assert key.unique().numel() == key.numel()
    adj = adjs[0]
    x = torch.randn(adj.size[0], 128, generator=g, device=dev)
    msg = x[adj.edge_index[0]]
    got = eng.c_segment_mean(msg, adj.edge_index[1], adj.size[1])       # the adopted plan (no sort)
    stats0 = dict(eng.stats)
    eng.seg_cache.clear()
    ref = eng.c_segment_mean(msg, adj.edge_index[1].clone(), adj.size[1])  # fresh plan from the ids
    assert eng.stats["plans_built"] == stats0["plans_built"] + 1 and torch.equal(got, ref)
    sage = layers.SAGEConv(128, 64, activation=torch.relu).to(dev)
    y = sage((x, x[: adj.size[1]]), adj.edge_index)
    hs = x @ sage.fc_neigh.weight.t()
    cnt = (adj.rowptr[1:] - adj.rowptr[:-1]).clamp(min=1).unsqueeze(1)
    formula = torch.zeros(adj.size[1], 64, device=dev).index_add_(0, adj.edge_index[1], hs[adj.edge_index[0]]) / cnt
    formula = torch.relu(formula + x[: adj.size[1]] @ sage.fc_self.weight.t() + sage.bias)
    torch.testing.assert_close(y, formula, rtol=1e-4, atol=1e-4)
    # independence across calls: the same seeds sampled again share few picks (the seeds' block draws 25 of
    # >= 100 neighbours: expected overlap 25 / deg <= 0.25; the first Philox layout gave ~6x the expectation)
    _, _, adjs2 = ns.sample(seeds)
    a, b = adjs[1], adjs2[1]
    rows = torch.nonzero(deg[seeds] >= 100).reshape(-1)[:256]
    assert rows.numel() >= 32
    same = 0
    for r in rows.tolist():
        s1 = set(a.e_id[int(a.rowptr[r]):int(a.rowptr[r + 1])].tolist())
        s2 = set(b.e_id[int(b.rowptr[r]):int(b.rowptr[r + 1])].tolist())
        assert len(s1) == 25 and len(s2) == 25
        same += len(s1 & s2)
    assert same / (25 * len(rows)) < 0.3, same / (25 * len(rows))


def test_one_rank_share_of_an_8_way_partition(eng, dev):
    """Config 5's construction path on one GPU at the products size: rank 3 of an 8-way partition built by the
    per-rank generator (dry partition: send lists and buffers as in a real 8-rank run, nothing on the wire).  With
    the halo buffer filled from the global activation the share's aggregate == the same rows of the global SpMM."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd.dist import PartitionedGraph, build_partition
    from gammagl_amd.synth import DATASETS, rmat_partitioned

    n, e, _, _ = DATASETS["products"]
    P, r = 8, 3
    pg = build_partition(n, e, 0, r, 1, None, dev, eng, parts=P)
    assert pg.dry and pg.e_global == e + n and 0.8 * pg.e_global / P < pg.e_local < 1.25 * pg.e_global / P
    assert pg.n_halo > 0 and len(pg.send_splits) == P and pg.send_splits[r] == 0 and pg.n_send > 0
    full = rmat_partitioned(n, e, seed=0, device=dev)                  # the same graph, whole
    ei = torch.stack([full["src"], full["dst"]])
    K = 64
    h = torch.randn(n, K, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    want = eng.c_spmm_sum(ei, full["w"], h)[pg.lo:pg.hi]

    class _Filled:
        def wait(self):
            return True

    def fill(out_rows, inp, out_splits, in_splits, tag=None):           # what the 7 peers would have sent
        assert out_rows == pg.n_halo and inp.shape[0] == pg.n_send
        return h[pg.halo_ids].contiguous(), _Filled()

    pg._a2a = fill
    from gammagl_amd import dist as gd
    old = gd.HALO_CHUNKS
    gd.HALO_CHUNKS = 1
    try:
        got = pg.aggregate(h[pg.lo:pg.hi].contiguous())
    finally:
        gd.HALO_CHUNKS = old
    bound = eng.c_spmm_sum(ei, full["w"], h.abs())[pg.lo:pg.hi]
    assert bool(((got - want).abs() <= 1e-5 * bound + 1e-6).all())
    # the rows this rank would send are exactly the rows of its range that other parts' edges read
    src, dst = full["src"], full["dst"]
    mine_src = (src >= pg.lo) & (src < pg.hi) & ((dst < pg.lo) | (dst >= pg.hi))
    assert torch.equal(torch.unique(pg.send_idx), torch.unique(src[mine_src]) - pg.lo)
    assert isinstance(pg, PartitionedGraph)


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
def test_arxiv_size_16bit_sums_with_a_reddit_sized_hub_row(eng, dev, oracle, dt):
    """f16 / bf16 unsorted_segment_sum / mean at config 2's size with a 109 110-element hub row (the Reddit-sized
    graph's longest): sums that accumulate in the storage type depend on the serial order far beyond rounding, and the
    hub rows go through the LDS-pipelined kernel (hub16.hip) — bit for bit the oracle's result, and the row walk's."""
    import numpy as np

    import parity_cases as pc
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["arxiv"]
    ids = rmat_graph(n, e, seed=0, device=dev)[1].contiguous()
    ids[:109110] = 5
    g = torch.Generator(device=dev).manual_seed(13)
    for K in (16, 7, 47):     # (47: ragged 8-wide lanes + the hub kernel's 16-byte producers for rows that are not 16-byte pieces)
        xf = (torch.randn(ids.shape[0], K, generator=g, device=dev) * 3 + 0.5).cpu().numpy()
        xh = oracle.f32_to_bf16_bits(xf) if dt == "bfloat16" else xf.astype(np.float16)
        xt = pc.to_t(xh, dev, dt)
        ids_h = ids.cpu().numpy()
        want_s = oracle.segment_sum(xh, ids_h, n, bf16=dt == "bfloat16")
        want_m = oracle.segment_mean(xh, ids_h, n, bf16=dt == "bfloat16")
        assert eng.seg_plan(ids, n).n_long >= 1
        for hub16 in (True, False):
            eng.hub16 = hub16
            try:
                pc.assert_same(pc.to_np(eng.c_segment_sum(xt, ids, n)), want_s, f"{dt} K{K} sum hub16={hub16}")
                pc.assert_same(pc.to_np(eng.c_segment_mean(xt, ids, n)), want_m, f"{dt} K{K} mean hub16={hub16}")
            finally:
                eng.hub16 = True


def test_products_size_column_blocks_same_bits(eng, dev):
    """The K = 256 aggregate as 4 launches over 64-column blocks (reduce.hip launch_f32_cols) == one launch over the
    1 KiB rows, bit for bit, at the products size: plain sum, its transpose, mean, and the fused epilogue with bias,
    ReLU and dropout (the blocks draw the full-width mask: epi_K / epi_col0)."""
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    w = torch.rand(ei.shape[1], generator=g, device=dev)
    gp = eng.graph_plan(ei, n)
    K = 256
    x = torch.randn(n, K, generator=g, device=dev)
    bias = torch.randn(K, generator=g, device=dev)
    assert eng.lib.ggl_spmm_col_blocks(gp.E, K, n) == 4 and eng.lib.ggl_spmm_col_blocks(gp.E, 100, n) == 1
    assert eng.lib.ggl_spmm_col_blocks(gp.E, K, gp.E // 5) == 1      # 5 edges per row: nothing to keep in L2
    old = eng.lib.ggl_get_option(b"col_block")
    res = {}
    try:
        for cb in (0, 64):
            eng.set_option("col_block", cb)
            outs = []
            a = torch.empty(n, K, device=dev)
            eng.spmm_sum_into(gp.fwd, gp.col, w, x, a)
            outs.append(a.clone())
            eng.spmm_sum_into(gp.bwd, gp.colT, w, x, a)
            outs.append(a.clone())
            eng.spmm_epi_into(gp.fwd, gp.col, w, x, a, mean=True, epi_K=K)
            outs.append(a.clone())
            eng.reseed(77)
            rng = eng._rng_state(dev)
            st = rng.clone()
            eng.spmm_epi_into(gp.fwd, gp.col, w, x, a, bias=bias, relu=True, p_drop=0.5, rng=rng, epi_K=K)
            assert int(rng[1]) != int(st[1])          # the state advanced once for the whole aggregate
            outs.append(a.clone())
            res[cb] = outs
    finally:
        eng.set_option("col_block", old)
    for a, b, nm in zip(res[0], res[64], ("sum", "transposed sum", "mean", "bias+relu+dropout")):
        assert torch.equal(a, b), nm
    kept = (res[64][3] != 0).float().mean()
    assert 0.2 < float(kept) < 0.3                    # relu x dropout(0.5) of a zero-mean aggregate


def test_locality_ordered_graph_gets_xcd_runs_same_bits(eng, dev):
    """Products-sized planted-community graph: in its own node order the plan detects locality and hands each XCD runs of
    2048 consecutive rows, with shuffled ids it does not; either way the aggregate's bits are the ones of the round-robin
    hand-out (scheduling only), forward and transposed."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd.synth import DATASETS, planted_pairs

    n, e, _, _ = DATASETS["products"]
    s_, d_ = planted_pairs(n, out_deg=max(2, e // (2 * n)), seed=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(n, 64, generator=g, device=dev)
    old = eng.xcd_run_rows
    try:
        for shuffled in (False, True):
            if shuffled:
                pi = torch.randperm(n, generator=g, device=dev)
                ei = torch.stack([pi[s_], pi[d_]]).contiguous()
            else:
                ei = torch.stack([s_, d_]).contiguous()
            res = {}
            for run in (-1, 0):
                eng.clear_caches()
                eng.xcd_run_rows = run
                gp = eng.graph_plan(ei, n)
                if run < 0:
                    loc = gp.locality()
                    assert (loc < 0.1 and gp.fwd.xcd_run == 0) if shuffled else (loc > 0.5 and gp.fwd.xcd_run == 2048), loc
                    assert gp.bwd.xcd_run == gp.fwd.xcd_run
                a, b = torch.empty(n, 64, device=dev), torch.empty(n, 64, device=dev)
                eng.spmm_sum_into(gp.fwd, gp.col, None, x, a)
                eng.spmm_sum_into(gp.bwd, gp.colT, None, x, b)
                res[run] = (a, b)
            assert torch.equal(res[-1][0], res[0][0]) and torch.equal(res[-1][1], res[0][1])
            del ei
    finally:
        eng.xcd_run_rows = old
        eng.clear_caches()


def test_products_size_bspmm_weight_gradient_sorted_plan_same_bits(eng, dev):
    """bspmm's weight gradient at the products size: the walk along the sorted plan (LDS-staged strips, 64-column-block
    launches carrying the running dot) == the thread-per-item kernel in edge order, bit for bit, for a one-head 256-channel
    row (4 column blocks) and 8 heads x 44 channels (one 32-column slab + a 12-column tail), and both == an f64 evaluation of a
    row sample within f32 rounding."""
    if not _big(dev):
        pytest.skip("needs > 100 GB of HBM")
    from gammagl_amd.synth import DATASETS, rmat_graph

    n, e, _, _ = DATASETS["products"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    E = ei.shape[1]
    g = torch.Generator(device=dev).manual_seed(4)
    for H, C in ((1, 256), (8, 44)):
        x = torch.randn(n, H, C, generator=g, device=dev).requires_grad_(True)
        w = torch.rand(E, H, generator=g, device=dev).requires_grad_(True)
        go = torch.randn(n, H, C, generator=g, device=dev)
        assert (eng.lib.ggl_bspmm_grad_w_sorted_scratch_bytes(E, n, H, C) > 0) == (C >= 128)
        got = {}
        for mode in (True, False):
            eng.gradw_sorted = mode
            x.grad = w.grad = None
            eng.c_bspmm_sum(ei, w, x).backward(go)
            got[mode] = w.grad.clone()
        eng.gradw_sorted = True
        assert torch.equal(got[True], got[False])
        pick = torch.randint(0, E, (4096,), generator=g, device=dev)
        want = (x.detach()[ei[0, pick]].double() * go[ei[1, pick]].double()).sum(-1)
        scale = (x.detach()[ei[0, pick]].abs().double() * go[ei[1, pick]].abs().double()).sum(-1)
        assert bool(((got[True][pick].double() - want).abs() <= 1e-5 * scale + 1e-6).all())
        del x, w, go, got
