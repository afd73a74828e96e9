"""Parity checks shared by the GPU suite (tests/test_gpu_parity.py: the HIP library on a real
MI355X, through the C ABI) and the CPU logic suite (tests/test_emul_kernels.py: the same kernel
sources compiled for the host, see tests/emul/).  Every check compares an Engine against the
golden vectors the reference produced and/or against the oracle on seeded inputs.

Tolerances: integer/index work and argmax witnesses are bit-exact; float reductions are bit-exact
wherever a row is reduced in one piece (same order of the same rounded operations as the serial
reference) and within 1e-5 relative (north_star) where long rows are split into chunks.
"""
import numpy as np
import torch

DT = {"uint8": torch.uint8, "int8": torch.int8, "int16": torch.int16, "int32": torch.int32,
      "int64": torch.int64, "float16": torch.float16, "bfloat16": torch.bfloat16,
      "float32": torch.float32, "float64": torch.float64}
KAT_DT = ["int8", "int16", "int32", "int64", "float16", "float32", "float64"]


class option:
    """`with option(eng, "col_block_min_degree", 0): ...` — a library option for the duration of a block"""

    def __init__(self, eng, name, value):
        self.eng, self.name, self.value = eng, name, value

    def __enter__(self):
        self.old = int(self.eng.lib.ggl_get_option(self.name.encode()))
        self.eng.set_option(self.name, self.value)

    def __exit__(self, *exc):
        self.eng.set_option(self.name, self.old)


def to_t(a, dev, dt=None):
    a = np.asarray(a)
    if dt == "bfloat16":
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(dev)
    return torch.from_numpy(a.copy()).to(dev)


def to_np(t):
    t = t.detach().cpu()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        ok = ((a == b) | (np.isnan(a) & np.isnan(b))) & (np.signbit(a) == np.signbit(b))
        return bool(ok.all())
    return bool((a == b).all())


def assert_same(a, b, what=""):
    assert same(a, b), f"{what}: max|diff|={np.nanmax(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))}"


# --------------------------------------------------------------------------------------------------
def check_kat(eng, dev, golden):
    g = golden["kat"]
    idx = to_t(g["idx"], dev)
    for op, fn in (("sum", eng.c_segment_sum), ("mean", eng.c_segment_mean), ("max", eng.c_segment_max)):
        for dt in KAT_DT:
            for dim in (1, 2, 3):
                x = to_t(g[f"{op}_{dt}_d{dim}_x"], dev)
                assert_same(to_np(fn(x, idx, 2)), g[f"{op}_{dt}_d{dim}_y"], f"kat {op} {dt} d{dim}")
    ei = to_t(g["mp_ei"], dev)
    msg = to_t(g["mp_x"], dev)[ei[0]]
    assert_same(to_np(eng.c_segment_sum(msg, ei[1].contiguous(), 4)), g["mp_sum"], "mp sum")
    assert_same(to_np(eng.c_segment_mean(msg, ei[1].contiguous(), 4)), g["mp_mean"], "mp mean")
    assert_same(to_np(eng.c_segment_max(msg, ei[1].contiguous(), 4)), g["mp_max"], "mp max")
    ones = torch.ones(5, dtype=torch.int64, device=dev)
    assert_same(to_np(eng.c_segment_sum(ones, to_t(g["deg_row"], dev), 3)), g["deg_out"], "degree")
    assert_same(to_np(eng.c_segment_max(to_t(g["doc_x"], dev), to_t(g["doc_ids"], dev), 3)), g["doc_max"], "doc max")
    y = eng.c_spmm_sum(to_t(g["doc_gi"], dev), 2 * torch.ones(8, device=dev), 2 * torch.ones(5, 8, device=dev))
    assert_same(to_np(y), g["doc_gspmm"], "doc gspmm")
    # segment_softmax composed exactly like utils/softmax.py:29-35
    x_e = to_t(g["sm_x"], dev)
    dst = ei[1].contiguous()
    mx = eng.c_segment_max(x_e, dst, 4)
    ex = torch.exp(x_e - mx[dst])
    den = eng.c_segment_sum(ex, dst, 4)
    score = ex / (den[dst] + 1e-16)
    np.testing.assert_allclose(to_np(score), g["sm_score"], rtol=1e-6)


def check_segment_all_dtypes(eng, dev, golden):
    g = golden["segment"]
    for ci in range(int(g["ncases"])):
        ids = to_t(g[f"c{ci}_ids"], dev)
        N = int(g[f"c{ci}_N"])
        for dt in DT:
            x = to_t(g[f"c{ci}_{dt}_x"], dev, dt)
            assert_same(to_np(eng.c_segment_sum(x, ids, N)), g[f"c{ci}_{dt}_sum"], f"c{ci} {dt} sum")
            assert_same(to_np(eng.c_segment_mean(x, ids, N)), g[f"c{ci}_{dt}_mean"], f"c{ci} {dt} mean")
            assert_same(to_np(eng.c_segment_max(x, ids, N)), g[f"c{ci}_{dt}_max"], f"c{ci} {dt} max")


def check_segment_fwd_bwd(eng, dev, golden):
    g = golden["segment"]
    for bi in range(int(g["nbwd"])):
        ids = to_t(g[f"b{bi}_ids"], dev)
        N = int(g[f"b{bi}_N"])
        for dt in ("float32", "float64"):
            k = f"b{bi}_{dt}"
            go = to_t(g[k + "_g"], dev)
            for name, fn in (("sum", eng.c_segment_sum), ("mean", eng.c_segment_mean), ("max", eng.c_segment_max)):
                x = to_t(g[k + "_x"], dev).requires_grad_(True)
                y = fn(x, ids, N)
                y.backward(go)
                assert_same(to_np(y), g[f"{k}_{name}"], f"{k} {name}")
                # for max the gradient is the argmax witness: bit-exact <=> every argmax agrees
                assert_same(to_np(x.grad), g[f"{k}_{name}_gx"], f"{k} {name} grad")


def check_special_values(eng, dev, golden):
    g = golden["segment"]
    ids = to_t(g["sp_ids"], dev)
    x = to_t(g["sp_x"], dev).requires_grad_(True)
    y = eng.c_segment_max(x, ids, 4)
    y.backward(torch.arange(16, dtype=torch.float32, device=dev).reshape(4, 4) + 1)
    assert_same(to_np(y), g["sp_max"], "nan/inf/-0 max")
    assert_same(to_np(x.grad), g["sp_gx"], "nan/inf/-0 max grad")
    assert_same(to_np(eng.c_segment_sum(x.detach(), ids, 4)), g["sp_sum"], "nan/inf sum")
    sid = to_t(g["sat_ids"], dev)
    for nm, dt in (("f16", "float16"), ("bf16", "bfloat16")):
        xs = to_t(g[f"sat_{nm}_x"], dev, dt)
        assert_same(to_np(eng.c_segment_sum(xs, sid, 3)), g[f"sat_{nm}_sum"], nm + " sum saturation")
        assert_same(to_np(eng.c_segment_mean(xs, sid, 3)), g[f"sat_{nm}_mean"], nm + " count saturation")


def check_spmm_golden(eng, dev, golden):
    g = golden["spmm"]
    for ci in range(int(g["nspmm"])):
        k = f"s{ci}"
        idx, w, go = to_t(g[k + "_index"], dev), to_t(g[k + "_w"], dev), to_t(g[k + "_g"], dev)
        for red, fn in (("sum", eng.c_spmm_sum), ("mean", eng.c_spmm_mean), ("max", eng.c_spmm_max)):
            x = to_t(g[k + "_x"], dev).requires_grad_(True)
            y = fn(idx, w, x)
            assert_same(to_np(y), g[f"{k}_{red}"], f"{k} {red}")
            if idx.shape[1] > 0:
                y.backward(go)
                assert_same(to_np(x.grad), g[f"{k}_{red}_gx"], f"{k} {red} grad")
    for bi in range(int(g["nbspmm"])):
        k = f"bs{bi}"
        idx, go = to_t(g[k + "_index"], dev), to_t(g[k + "_g"], dev)
        w = to_t(g[k + "_w"], dev).requires_grad_(True)
        x = to_t(g[k + "_x"], dev).requires_grad_(True)
        y = eng.c_bspmm_sum(idx, w, x)
        y.backward(go)
        assert_same(to_np(y), g[k + "_y"], k)
        assert_same(to_np(x.grad), g[k + "_gx"], k + " gx")
        assert_same(to_np(w.grad), g[k + "_gw"], k + " gw")


def check_layers_golden(eng, dev, golden):
    g = golden["layers"]
    ei = to_t(g["gcn_ei"], dev)
    x, b = to_t(g["gcn_x"], dev), to_t(g["gcn_b"], dev)
    W = to_t(g["gcn_W"], dev).requires_grad_(True)
    N = x.shape[0]
    src, dst = ei[0].contiguous(), ei[1].contiguous()
    ones = torch.ones(ei.shape[1], device=dev)
    wts = eng.c_segment_sum(ones, src, N).pow(-0.5)[src] * ones
    wts = wts * eng.c_segment_sum(ones, dst, N).pow(-0.5)[dst]
    np.testing.assert_allclose(to_np(wts), g["gcn_w"], rtol=1e-6)
    # unfused, as GCNConv runs today: gather * w -> unsorted_segment_sum (message_passing.py:56-59,84-86)
    h = x @ W
    y = eng.c_segment_sum(h[src] * wts.unsqueeze(-1), dst, N) + b
    y.backward(to_t(g["gcn_g"], dev))
    np.testing.assert_allclose(to_np(y), g["gcn_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(to_np(W.grad), g["gcn_gW"], rtol=1e-4, atol=1e-5)
    # fused: gspmm
    W2 = to_t(g["gcn_W"], dev).requires_grad_(True)
    y2 = eng.c_spmm_sum(ei, wts.detach(), x @ W2) + b
    y2.backward(to_t(g["gcn_g"], dev))
    np.testing.assert_allclose(to_np(y2), g["gcn_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(to_np(W2.grad), g["gcn_gW"], rtol=1e-4, atol=1e-5)

    # GAT: fused kernel vs the reference's unfused chain (gat_conv.py:103-112)
    gei = to_t(g["gat_ei"], dev)
    att = to_t(g["gat_att"], dev)
    xg = to_t(g["gat_x"], dev).requires_grad_(True)
    C = xg.shape[2]
    el = (xg * att[:, :, :C]).sum(-1)
    er = (xg * att[:, :, C:]).sum(-1)
    yg = eng.gat_fused(gei, el, er, xg, 0.2)
    yg.backward(to_t(g["gat_g"], dev))
    np.testing.assert_allclose(to_np(yg), g["gat_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(to_np(xg.grad), g["gat_gx"], rtol=2e-4, atol=2e-5)
    check_headmean_golden(eng, dev, golden)


def _row_scale_err(got, want):
    """max |got - want| / max(|want|, the row's largest |want|): oracle/parity.py's criterion on numpy arrays."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    g2 = got.reshape(got.shape[0], -1) if got.ndim > 1 else got.reshape(1, -1)
    w2 = want.reshape(g2.shape)
    floor = np.abs(w2).max(axis=1, keepdims=True).clip(min=1e-30)
    return float((np.abs(g2 - w2) / np.maximum(np.abs(w2), floor)).max())


def check_headmean_golden(eng, dev, golden, tol=1e-5, gtol=2e-5):
    """The head-averaging output layer (gat_conv.py:98-122 with concat=False) and the two-layer GATModel (models/gat.py)
    through FusedGATConv / GATModel(fused=True) against vectors the REFERENCE's segment ops produced (layers.npz gatm_*,
    gatmodel_*): on the GPU FusedGATConv(concat=False) runs the ggl_gat_sh_* kernels, on the host build the generic fused path."""
    import gammagl_amd
    from gammagl_amd.layers import FusedGATConv, GATModel

    g = golden["layers"]
    ei = to_t(g["gat_ei"], dev)
    N = int(g["gatm_x"].shape[0])
    H = int(g["gatm_att"].shape[1])
    C = int(g["gatm_att"].shape[2]) // 2
    prev = gammagl_amd._engine
    gammagl_amd._engine = eng            # the layers resolve their engine through the package
    try:
        layer = FusedGATConv(int(g["gatm_x"].shape[1]), C, heads=H, concat=False).to(dev)
        with torch.no_grad():
            layer.w.copy_(to_t(g["gatm_W"], dev)), layer.att.copy_(to_t(g["gatm_att"], dev)), layer.bias.copy_(to_t(g["gatm_b"], dev))
        x = to_t(g["gatm_x"], dev).requires_grad_(True)
        y = layer(x, ei, N)
        y.backward(to_t(g["gatm_g"], dev))
        assert _row_scale_err(to_np(y), g["gatm_y"]) <= tol
        for got, key in ((x.grad, "gx"), (layer.w.grad, "gW"), (layer.att.grad, "gatt"), (layer.bias.grad, "gb")):
            err = _row_scale_err(to_np(got), g["gatm_" + key])
            assert err <= gtol, (key, err)
        Hd = int(g["gatmodel_att0"].shape[2]) // 2
        NC = int(g["gatmodel_att1"].shape[2]) // 2
        model = GATModel(int(g["gatmodel_x"].shape[1]), Hd, NC, heads=H, drop_rate=0.0, num_layers=2, fused=True).to(dev).eval()
        with torch.no_grad():
            for li, conv in enumerate(model.gat_list):
                conv.w.copy_(to_t(g[f"gatmodel_W{li}"], dev)), conv.att.copy_(to_t(g[f"gatmodel_att{li}"], dev))
                conv.bias.copy_(to_t(g[f"gatmodel_b{li}"], dev))
        y2 = model(to_t(g["gatmodel_x"], dev), ei, N)
        y2.backward(to_t(g["gatmodel_g"], dev))
        assert _row_scale_err(to_np(y2), g["gatmodel_y"]) <= tol
        for li, conv in enumerate(model.gat_list):
            for got, key in ((conv.w.grad, "gW"), (conv.att.grad, "gatt"), (conv.bias.grad, "gb")):
                err = _row_scale_err(to_np(got), g[f"gatmodel_{key}{li}"])
                assert err <= gtol, (li, key, err)
    finally:
        gammagl_amd._engine = prev


# --------------------------------------------------------------------------------------------------
def _rand_graph(rng, N, E, hub=None):
    src = rng.integers(0, N, size=E)
    dst = rng.integers(0, N, size=E)
    if hub:
        dst[: E // 3] = hub
    return np.stack([src, dst]).astype(np.int64)


def check_random_vs_oracle(eng, dev, oracle, seed=0, sizes=None):
    """segment ops + gspmm + bspmm on seeded graphs vs the oracle, many K, sorted and unsorted ids."""
    rng = np.random.default_rng(seed)
    sizes = sizes or [(50, 400), (257, 3000)]
    for (N, E) in sizes:
        for K in (1, 3, 4, 8, 16, 47, 64, 100, 256, 260):
            ids = rng.integers(0, N, size=E).astype(np.int64)
            if K in (8, 256):
                ids.sort()  # already-sorted fast path (no perm)
            x = rng.standard_normal((E, K)).astype(np.float32)
            xt, it = to_t(x, dev), to_t(ids, dev)
            assert_same(to_np(eng.c_segment_sum(xt, it, N)), oracle.segment_sum(x, ids, N), f"sum N{N} K{K}")
            assert_same(to_np(eng.c_segment_mean(xt, it, N)), oracle.segment_mean(x, ids, N), f"mean N{N} K{K}")
            mx, arg = eng.segment_max_with_arg(xt, it, N)
            omx, oarg = oracle.segment_max(x, ids, N)
            assert_same(to_np(mx), omx, f"max N{N} K{K}")
            assert_same(to_np(arg), oarg, f"argmax N{N} K{K}")
        for K in (1, 4, 7, 16, 47, 64, 256):
            index = _rand_graph(rng, N, E)
            if K == 16:
                index = index[:, np.argsort(index[1], kind="stable")]  # CSR-ordered edge list
            w = rng.standard_normal(E).astype(np.float32)
            x = rng.standard_normal((N, K)).astype(np.float32)
            go = rng.standard_normal((N, K)).astype(np.float32)
            it, wt, gt = to_t(index, dev), to_t(w, dev), to_t(go, dev)
            for red, fn in (("sum", eng.c_spmm_sum), ("mean", eng.c_spmm_mean), ("max", eng.c_spmm_max)):
                xt = to_t(x, dev).requires_grad_(True)
                y = fn(it, wt, xt)
                y.backward(gt)
                if red == "sum":
                    oy, ogx = oracle.spmm_sum_fwd(index, w, x), oracle.spmm_sum_bwd(index, w, go)
                elif red == "mean":
                    oy, cnt = oracle.spmm_mean_fwd(index, w, x)
                    ogx = oracle.spmm_mean_bwd(index, w, go, cnt)
                else:
                    oy, arg = oracle.spmm_max_fwd(index, w, x)
                    ogx = oracle.spmm_max_bwd(index, w, go, arg)
                assert_same(to_np(y), oy, f"spmm {red} N{N} K{K}")
                assert_same(to_np(xt.grad), ogx, f"spmm {red} grad N{N} K{K}")
            # weight=None == ones
            y1 = eng.c_spmm_sum(it, None, to_t(x, dev))
            assert_same(to_np(y1), oracle.spmm_sum_fwd(index, np.ones(E, np.float32), x), "spmm w=None")
        for (H, C) in ((8, 8), (4, 16), (3, 5), (1, 64), (8, 32)):
            index = _rand_graph(rng, N, E)
            w = rng.standard_normal((E, H)).astype(np.float32)
            x = rng.standard_normal((N, H, C)).astype(np.float32)
            go = rng.standard_normal((N, H, C)).astype(np.float32)
            wt = to_t(w, dev).requires_grad_(True)
            xt = to_t(x, dev).requires_grad_(True)
            y = eng.c_bspmm_sum(to_t(index, dev), wt, xt)
            y.backward(to_t(go, dev))
            ogx, ogw = oracle.bspmm_sum_bwd(index, w, x, go)
            assert_same(to_np(y), oracle.bspmm_sum_fwd(index, w, x), f"bspmm H{H} C{C}")
            assert_same(to_np(xt.grad), ogx, f"bspmm gx H{H} C{C}")
            assert_same(to_np(wt.grad), ogw, f"bspmm gw H{H} C{C}")


def check_long_rows(eng, dev, oracle, chunk=8):
    """Force the chunked long-row path (threshold `chunk`) with hubs far longer than it."""
    old = eng.chunk
    eng.chunk = chunk
    eng.seg_cache.clear()
    eng.graph_cache.clear()
    try:
        rng = np.random.default_rng(5)
        N, E = 40, 1500
        for K in (1, 4, 5, 64, 256):
            ids = rng.integers(0, N, size=E).astype(np.int64)
            ids[:700] = 7
            ids[700:900] = 0
            ids[900:905] = 39
            # tie-prone values so that the chunk-ordered argmax tie-break is exercised
            x = (rng.integers(-3, 4, size=(E, K)) * 0.5).astype(np.float32)
            xt, it = to_t(x, dev), to_t(ids, dev)
            plan = eng.seg_plan(it, N)
            assert plan.n_long >= 2 and plan.n_chunks > plan.n_long
            np.testing.assert_allclose(to_np(eng.c_segment_sum(xt, it, N)), oracle.segment_sum(x, ids, N),
                                       rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(to_np(eng.c_segment_mean(xt, it, N)), oracle.segment_mean(x, ids, N),
                                       rtol=1e-5, atol=1e-5)
            mx, arg = eng.segment_max_with_arg(xt, it, N)
            omx, oarg = oracle.segment_max(x, ids, N)
            assert_same(to_np(mx), omx, f"long max K{K}")
            assert_same(to_np(arg), oarg, f"long argmax K{K}")
            # integers: wrap-around sums are order independent -> bit exact even when chunked
            xi = rng.integers(-100, 100, size=(E, K)).astype(np.int32)
            assert_same(to_np(eng.c_segment_sum(to_t(xi, dev), it, N)), oracle.segment_sum(xi, ids, N), "long i32")
            assert_same(to_np(eng.c_segment_mean(to_t(xi, dev), it, N)), oracle.segment_mean(xi, ids, N), "long i32 mean")
        # f16 / bf16 accumulate in the storage type: their hub rows are never chunked.  The GPU build
        # hands them to the LDS-pipelined hub kernel (hub16.hip; 16-byte pieces / element by element), in the emulated
        # build one lane group walks them: both must be the reference's serial result bit for bit — sums that saturate included.
        for K in (8, 40, 64, 72, 256, 5, 1, 47, 130):
            ids = rng.integers(0, N, size=E).astype(np.int64)
            ids[:700] = 7       # 700 > 2 x 256: several LDS stages + a ragged last one
            ids[700:1000] = 0
            ids[1000:1009] = 39
            xf = (rng.standard_normal((E, K)) * 40 + 3).astype(np.float32)
            it = to_t(ids, dev)
            for nm in ("float16", "bfloat16"):
                xh = oracle.f32_to_bf16_bits(xf) if nm == "bfloat16" else xf.astype(np.float16)
                xt = to_t(xh, dev, nm)
                want_s = oracle.segment_sum(xh, ids, N, bf16=nm == "bfloat16")
                want_m = oracle.segment_mean(xh, ids, N, bf16=nm == "bfloat16")
                for hub16 in (True, False):
                    eng.hub16 = hub16
                    assert_same(to_np(eng.c_segment_sum(xt, it, N)), want_s, f"{nm} hub sum K{K} hub16={hub16}")
                    assert_same(to_np(eng.c_segment_mean(xt, it, N)), want_m, f"{nm} hub mean K{K} hub16={hub16}")
                eng.hub16 = True
        for K in (4, 47, 256):
            index = _rand_graph(rng, N, E, hub=3)
            index[0, : E // 2] = 11  # a source hub too: long rows in the transposed plan
            w = rng.standard_normal(E).astype(np.float32)
            x = rng.standard_normal((N, K)).astype(np.float32)
            go = rng.standard_normal((N, K)).astype(np.float32)
            it, wt = to_t(index, dev), to_t(w, dev)
            gp = eng.graph_plan(it, N)
            assert gp.fwd.n_long >= 1 and gp.bwd.n_long >= 1
            xt = to_t(x, dev).requires_grad_(True)
            y = eng.c_spmm_sum(it, wt, xt)
            y.backward(to_t(go, dev))
            np.testing.assert_allclose(to_np(y), oracle.spmm_sum_fwd(index, w, x), rtol=1e-5, atol=1e-4)
            np.testing.assert_allclose(to_np(xt.grad), oracle.spmm_sum_bwd(index, w, go), rtol=1e-5, atol=1e-4)
            ym = eng.c_spmm_mean(it, wt, to_t(x, dev))
            np.testing.assert_allclose(to_np(ym), oracle.spmm_mean_fwd(index, w, x)[0], rtol=1e-5, atol=1e-5)
            xt2 = to_t(x, dev).requires_grad_(True)
            yx = eng.c_spmm_max(it, wt, xt2)
            yx.backward(to_t(go, dev))
            oyx, oarg = oracle.spmm_max_fwd(index, w, x)
            assert_same(to_np(yx), oyx, "long spmm max")
            np.testing.assert_allclose(to_np(xt2.grad), oracle.spmm_max_bwd(index, w, go, oarg), rtol=1e-5, atol=1e-4)
        # fused GAT with long rows in the transposed (source-major) backward
        H, C = 4, 8
        index = _rand_graph(rng, N, 600, hub=5)
        index[0, :300] = 2
        _check_gat(eng, dev, oracle, index, N, H, C, rng)
    finally:
        eng.chunk = old
        eng.seg_cache.clear()
        eng.graph_cache.clear()


def check_exact_long_rows(eng, dev, oracle, chunk=64):
    """f32 sums on rows LONGER than the plan's chunk are the reference's serial chain, add for add (hubf32.hip on the
    GPU: producer wavefronts gather, one consumer wavefront adds in element order; the host build walks rows in one
    piece): every mode that sums is BIT-IDENTICAL to the oracle on every row — real-valued inputs, where another
    association of the same adds shows in the last bits.  Hubs of several LDS stages, ragged last stages, a row just
    over the threshold, widths that are / are not 16-byte rows, column-block launches, strided + accumulate forms."""
    old = eng.chunk
    eng.chunk = chunk
    eng.clear_caches()
    try:
        rng = np.random.default_rng(17)
        N, E = 50, 5000
        hubs = ((7, 2900), (0, 1024), (33, 385), (12, chunk + 1))     # (row, length); stages are 128 elements

        def hub_ids():
            ids = rng.integers(0, N, size=E).astype(np.int64)
            at = 0
            for r, n in hubs:
                ids[at:at + n] = r
                at += n
            return ids

        # ---- segment sum / mean: unsorted ids (perm) and sorted ids
        for K in (1, 4, 5, 47, 64, 100, 130, 256):
            ids = hub_ids()
            rng.shuffle(ids)
            for sort in (False, True):
                ids_k = np.sort(ids) if sort else ids
                x = (rng.standard_normal((E, K)) * 3).astype(np.float32)
                xt, it = to_t(x, dev), to_t(ids_k, dev)
                plan = eng.seg_plan(it, N)
                assert plan.n_long >= 3
                assert_same(to_np(eng.c_segment_sum(xt, it, N)), oracle.segment_sum(x, ids_k, N), f"exact seg sum K{K} sorted={sort}")
                assert_same(to_np(eng.c_segment_mean(xt, it, N)), oracle.segment_mean(x, ids_k, N), f"exact seg mean K{K} sorted={sort}")
        # ---- ... and of doubles (the same pipeline moving the row as 4-byte words, the consumer adding doubles)
        for K in (1, 3, 4, 8, 9, 32, 33, 100):
            ids = hub_ids()
            rng.shuffle(ids)
            for sort in (False, True):
                ids_k = np.sort(ids) if sort else ids
                x = rng.standard_normal((E, K)) * 3
                xt, it = to_t(x, dev), to_t(ids_k, dev)
                assert_same(to_np(eng.c_segment_sum(xt, it, N)), oracle.segment_sum(x, ids_k, N), f"exact f64 seg sum K{K} sorted={sort}")
                assert_same(to_np(eng.c_segment_mean(xt, it, N)), oracle.segment_mean(x, ids_k, N), f"exact f64 seg mean K{K} sorted={sort}")
        # ---- gspmm sum / mean, forward and transposed backward; weights absent / first sight / sorted copy
        for K in (4, 48, 64, 100, 256):
            index = np.stack([rng.integers(0, N, size=E), hub_ids()]).astype(np.int64)
            index[0, E - 1500:] = 11                      # a source hub: long rows in the transposed plan too
            o = rng.permutation(E)
            index = np.ascontiguousarray(index[:, o])
            w = rng.standard_normal(E).astype(np.float32)
            x = rng.standard_normal((N, K)).astype(np.float32)
            go = rng.standard_normal((N, K)).astype(np.float32)
            it, wt = to_t(index, dev), to_t(w, dev)
            gp = eng.graph_plan(it, N)
            assert gp.fwd.n_long >= 3 and gp.bwd.n_long >= 1
            want, want_g = oracle.spmm_sum_fwd(index, w, x), oracle.spmm_sum_bwd(index, w, go)
            for call in range(3):                          # call 0: w[perm[p]] in the kernel; later: streamed sorted copy
                xt = to_t(x, dev).requires_grad_(True)
                y = eng.c_spmm_sum(it, wt, xt)
                y.backward(to_t(go, dev))
                assert_same(to_np(y), want, f"exact spmm sum K{K} call {call}")
                assert_same(to_np(xt.grad), want_g, f"exact spmm sum backward K{K} call {call}")
            assert_same(to_np(eng.c_spmm_mean(it, wt, to_t(x, dev))), oracle.spmm_mean_fwd(index, w, x)[0], f"exact spmm mean K{K}")
            ones = np.ones(E, np.float32)
            assert_same(to_np(eng.c_spmm_sum(it, None, to_t(x, dev))), oracle.spmm_sum_fwd(index, ones, x), f"exact spmm sum no weights K{K}")
            # the 64-column block launches (forced: the graph is far below the automatic threshold)
            if K % 64 == 0 and K >= 128:
                with option(eng, "col_block_min_edges", 0), option(eng, "col_block_min_degree", 0):
                    assert int(eng.lib.ggl_spmm_col_blocks(E, K, N)) > 1
                    # hub_one_launch = 1 (round 5): ONE hub launch over the full width in front of the first block + one
                    # long_final behind the last; 0: a hub launch and a long_final per column block — same bits
                    bb = rng.standard_normal(K).astype(np.float32)
                    acc0 = rng.standard_normal((N, K)).astype(np.float32)
                    for one in (1, 0):
                        with option(eng, "hub_one_launch", one):
                            xt = to_t(x, dev).requires_grad_(True)
                            y = eng.c_spmm_sum(it, wt, xt)
                            y.backward(to_t(go, dev))
                            assert_same(to_np(y), want, f"exact spmm sum K{K} column blocks one_hub={one}")
                            assert_same(to_np(xt.grad), want_g, f"exact spmm sum backward K{K} column blocks one_hub={one}")
                            assert_same(to_np(eng.c_spmm_mean(it, wt, to_t(x, dev))), oracle.spmm_mean_fwd(index, w, x)[0],
                                        f"exact spmm mean K{K} column blocks one_hub={one}")
                            ye = eng.spmm_epi(gp, wt, to_t(x, dev), "sum", bias=to_t(bb, dev), relu=True)
                            assert_same(to_np(ye), np.maximum(want + bb, 0).astype(np.float32),
                                        f"exact spmm epi K{K} column blocks one_hub={one}")
                            outa = to_t(acc0, dev)
                            eng.spmm_sum_into(gp.fwd, gp.col, wt, to_t(x, dev), outa, accumulate=True)
                            # (out += : the chain of a row starts from what `out` held — no oracle form; both launch
                            #  shapes must agree bit for bit and sit within rounding of the two-step sum)
                            got_acc = to_np(outa)
                            assert np.allclose(got_acc, acc0 + want, rtol=1e-4, atol=1e-3), f"accumulate K{K} one_hub={one}"
                            if one == 1:
                                first_acc = got_acc
                            else:
                                assert_same(got_acc, first_acc, f"accumulate K{K}: one hub launch vs one per block")
            # strided + accumulate: a column block of a wider matrix, a second edge set added onto a result
            if K >= 48 and K % 16 == 0:
                wide = to_t(np.concatenate([x, x[:, :16]], axis=1), dev)
                outw = torch.zeros(N, K + 16, device=dev)
                eng.spmm_sum_into(gp.fwd, gp.col, wt, wide[:, :K], outw[:, 16:])
                assert_same(to_np(outw[:, 16:].contiguous()), want, f"exact spmm_sum_into strided K{K}")
            # the fused epilogue rides on the same partial rows
            b = rng.standard_normal(K).astype(np.float32)
            if K % 4 == 0:
                ye = eng.spmm_epi(gp, wt, to_t(x, dev), "sum", bias=to_t(b, dev), relu=True)
                assert_same(to_np(ye), np.maximum(want + b, 0).astype(np.float32), f"exact spmm epi K{K}")
        # ---- bspmm: per-head weights
        for H, C in ((2, 8), (4, 16), (3, 5), (8, 32), (16, 16), (1, 256), (2, 128), (3, 128), (5, 24)):
            index = np.stack([rng.integers(0, N, size=E), hub_ids()]).astype(np.int64)
            index = np.ascontiguousarray(index[:, rng.permutation(E)])
            w = rng.standard_normal((E, H)).astype(np.float32)
            x = rng.standard_normal((N, H, C)).astype(np.float32)
            go = rng.standard_normal((N, H, C)).astype(np.float32)
            ogx, ogw = oracle.bspmm_sum_bwd(index, w, x, go)
            for call in range(2):
                wt, xt = to_t(w, dev).requires_grad_(True), to_t(x, dev).requires_grad_(True)
                y = eng.c_bspmm_sum(to_t(index, dev), wt, xt)
                y.backward(to_t(go, dev))
                assert_same(to_np(y), oracle.bspmm_sum_fwd(index, w, x), f"exact bspmm {H}x{C} call {call}")
                assert_same(to_np(xt.grad), ogx, f"exact bspmm gx {H}x{C} call {call}")
            # the 64-column-block launches (forced on this toy graph) where a block lies inside one head (1 x 256, 2 x 128,
            # 3 x 128 with its 384 columns); one launch where it would span heads (16 x 16, 8 x 32) or not nest (5 x 24)
            if H * C >= 128:
                with option(eng, "col_block_min_edges", 0), option(eng, "col_block_min_degree", 0):
                    wt, xt = to_t(w, dev).requires_grad_(True), to_t(x, dev).requires_grad_(True)
                    y = eng.c_bspmm_sum(to_t(index, dev), wt, xt)
                    y.backward(to_t(go, dev))
                    assert_same(to_np(y), oracle.bspmm_sum_fwd(index, w, x), f"exact bspmm {H}x{C} column blocks")
                    assert_same(to_np(xt.grad), ogx, f"exact bspmm gx {H}x{C} column blocks")
                    assert_same(to_np(wt.grad), ogw, f"exact bspmm gw {H}x{C} column blocks")
        # ---- the A/B switch: the chunked walk is still there, within rounding
        with option(eng, "exact_long_rows", 0):
            ids = hub_ids()
            x = (rng.standard_normal((E, 64)) * 3).astype(np.float32)
            np.testing.assert_allclose(to_np(eng.c_segment_sum(to_t(x, dev), to_t(ids, dev), N)),
                                       oracle.segment_sum(x, ids, N), rtol=1e-4, atol=1e-3)   # (another association of 2900 adds)
            chunked = to_np(eng.c_segment_sum(to_t(x, dev), to_t(ids, dev), N))
        # ---- ... and a plan whose LONGEST row exceeds `exact_long_max` takes it (a star graph's centre would be one add
        # chain of E elements): same bits as the switch above on the GPU; the host build never chunks a summing row
        longest = eng.seg_plan(to_t(ids, dev), N).max_len
        assert longest >= 2900
        with option(eng, "exact_long_max", longest - 1):
            capped = to_np(eng.c_segment_sum(to_t(x, dev), to_t(ids, dev), N))
        with option(eng, "exact_long_max", longest):
            exact = to_np(eng.c_segment_sum(to_t(x, dev), to_t(ids, dev), N))
        assert_same(exact, oracle.segment_sum(x, ids, N), "exact_long_max at the longest row")
        if dev != "cpu" and str(dev) != "cpu":
            assert_same(capped, chunked, "exact_long_max below the longest row = the chunked walk")
        else:
            assert_same(capped, exact, "host build: one piece either way")
    finally:
        eng.chunk = old
        eng.clear_caches()


def _check_gat(eng, dev, oracle, index, N, H, C, rng):
    el = rng.standard_normal((N, H)).astype(np.float32)
    er = rng.standard_normal((N, H)).astype(np.float32)
    x = rng.standard_normal((N, H, C)).astype(np.float32)
    go = rng.standard_normal((N, H, C)).astype(np.float32)
    elt, ert, xt = (to_t(a, dev).requires_grad_(True) for a in (el, er, x))
    y = eng.gat_fused(to_t(index, dev), elt, ert, xt, 0.2)
    y.backward(to_t(go, dev))
    oy = oracle.gat_fwd(index, el, er, x, 0.2)
    gel, ger, gx = oracle.gat_bwd(index, el, er, x, go, 0.2)
    np.testing.assert_allclose(to_np(y), oy, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(to_np(xt.grad), gx, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(to_np(elt.grad), gel, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(to_np(ert.grad), ger, rtol=1e-4, atol=2e-5)


def _check_gat_separate_buffers(eng, dev, index, N, H, C, rng):
    """C-ABI convention check: alpha / de as two [E,H] arrays give the same gradients as the interleaved
    [E,H,2] buffer the Engine passes (ggl_gat_fused_bwd_dst / _src accept both)."""
    import ctypes

    from gammagl_amd.ops import _ptr

    it = to_t(index, dev)
    el, er = (to_t(rng.standard_normal((N, H)).astype(np.float32), dev).requires_grad_(True) for _ in range(2))
    x = to_t(rng.standard_normal((N, H, C)).astype(np.float32), dev).requires_grad_(True)
    go = to_t(rng.standard_normal((N, H, C)).astype(np.float32), dev)
    fast_was = eng.gat_fast
    eng.gat_fast = False        # the Engine on the generic kernels: what the direct C-ABI calls below run
    try:
        y = eng.gat_fused(it, el, er, x, 0.2)
        y.backward(go)
    finally:
        eng.gat_fast = fast_was
    if fast_was and eng.lib.ggl_gat_fast_supported(H, C):
        # the fast kernels (v_exp_f32, block rescale, FMA, alpha / de recomputed in both backward walks) agree with
        # the generic ones to the GAT parity bar, forward and all three gradients
        el2, er2, x2 = (t.detach().clone().requires_grad_(True) for t in (el, er, x))
        y2 = eng.gat_fused(it, el2, er2, x2, 0.2)
        y2.backward(go)
        torch.testing.assert_close(y2.detach(), y.detach(), rtol=1e-5, atol=1e-5)
        for a, b in ((x2.grad, x.grad), (el2.grad, el.grad), (er2.grad, er.grad)):
            tol = 1e-4 * float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= tol, (H, C, float((a - b).abs().max()), tol)
    gp = eng.graph_plan(it, N)
    E = gp.E
    st = eng._stream(dev)
    out, rmax, rden = (torch.empty(s_, dtype=torch.float32, device=dev) for s_ in ((N, H, C), (N, H), (N, H)))
    part = None
    if gp.fwd.n_long > 0:
        part = torch.empty(eng.lib.ggl_gat_partial_bytes(gp.fwd.n_chunks, H, C) + 16, dtype=torch.uint8, device=dev)
    cs = gp.fwd.c_struct(part)
    eld, erd, xd = el.detach(), er.detach(), x.detach()
    eng._check(eng.lib.ggl_gat_fused_fwd(ctypes.byref(cs), _ptr(gp.col), _ptr(eld), _ptr(erd), _ptr(xd), 0.2, H, C,
                                         0.0, None, _ptr(out), _ptr(rmax), _ptr(rden), st))
    alpha = torch.empty((max(E, 1), H), dtype=torch.float32, device=dev)
    de = torch.empty((max(E, 1), H), dtype=torch.float32, device=dev)
    ger = torch.empty((N, H), dtype=torch.float32, device=dev)
    pf = eng._partial(gp.fwd, torch.float32, H, False, dev)
    cs = gp.fwd.c_struct(pf)
    eng._check(eng.lib.ggl_gat_fused_bwd_dst(ctypes.byref(cs), _ptr(gp.col), None, _ptr(eld), _ptr(erd), _ptr(xd),
                                             _ptr(go), _ptr(out), _ptr(rmax), _ptr(rden), 0.2, H, C, 0.0, None,
                                             _ptr(alpha), _ptr(de), _ptr(ger), None, st))
    gx = torch.empty((N, H, C), dtype=torch.float32, device=dev)
    gel = torch.empty((N, H), dtype=torch.float32, device=dev)
    pb = eng._partial(gp.bwd, torch.float32, H * C + H, False, dev)
    csT = gp.bwd.c_struct(pb)
    eng._check(eng.lib.ggl_gat_fused_bwd_src(ctypes.byref(csT), _ptr(gp.colT), _ptr(gp.posT), _ptr(alpha), _ptr(de),
                                             _ptr(go), H, C, _ptr(gx), _ptr(gel), st))
    assert torch.equal(out, y.detach())
    assert torch.equal(ger, er.grad) and torch.equal(gel, el.grad) and torch.equal(gx, x.grad)  # deterministic either way


def philox4x32_10(index, offset, seed):
    """numpy restatement of the device generator (csrc/common.hpp): the four words of Philox4x32-10 with
    counter (index, offset) and key seed — used to rebuild the attention-dropout mask on the host."""
    index = np.asarray(index, dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    c0, c1 = index & m32, index >> np.uint64(32)
    c2 = np.full_like(index, np.uint64(offset) & m32)
    c3 = np.full_like(index, np.uint64(offset) >> np.uint64(32))
    k0, k1 = np.uint64(seed) & m32, np.uint64(seed) >> np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & m32
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & m32
        c0, c1, c2, c3 = n0 & m32, n1, n2 & m32, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def gat_drop_word(index, offset, seed):
    """numpy restatement of gat_common.hpp drop_word: the attention-dropout word of counter index = p * H + h."""
    m64 = (1 << 64) - 1
    z = (int(seed) + int(offset) * 0x9E3779B97F4A7C15) & m64        # drop_key: splitmix64 of the launch's (seed, offset)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
    z ^= z >> 31
    k0, k1 = np.uint64(z & 0xFFFFFFFF), np.uint64(z >> 32)
    idx = np.asarray(index, dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    lo, hi = idx & m32, idx >> np.uint64(32)
    x = ((lo ^ k0) * np.uint64(0x9E3779B1)) & m32
    x ^= x >> np.uint64(15)
    x = ((x ^ hi ^ k1) * np.uint64(0x85EBCA77)) & m32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE3D)) & m32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def check_drop_word_statistics():
    """The attention-dropout generator as a random source (host restatement; the device kernels are held to it word
    for word by check_gat_dropout): keep rate at the thresholds the layers use, no correlation between the masks of
    consecutive steps (offset, offset + 1) or of neighbouring indices, every output bit balanced, indices that differ
    only above bit 32 get different words."""
    n = 1 << 20
    idx = np.arange(n, dtype=np.uint64)
    tol = 5.0 / np.sqrt(n)
    for seed in (0, 1, 0x123456789ABCDEF, (1 << 62) - 3):
        w0 = gat_drop_word(idx, 7, seed)
        w1 = gat_drop_word(idx, 8, seed)
        for pd in (0.1, 0.5, 0.6):
            t = np.uint32(int(pd * 4294967296.0))
            k0, k1 = (w0 >= t).astype(np.float64), (w1 >= t).astype(np.float64)
            assert abs(k0.mean() - (1 - pd)) < tol and abs(k1.mean() - (1 - pd)) < tol, (seed, pd, k0.mean())
            c_steps = np.corrcoef(k0, k1)[0, 1]                      # step n vs step n + 1, same positions
            c_neigh = np.corrcoef(k0[:-1], k0[1:])[0, 1]              # position p vs p + 1, same step
            c_head = np.corrcoef(k0[:-8], k0[8:])[0, 1]               # (p, h) vs (p + 1, h) with 8 heads
            assert abs(c_steps) < tol and abs(c_neigh) < tol and abs(c_head) < tol, (seed, pd, c_steps, c_neigh, c_head)
        bits = ((w0[:, None] >> np.arange(32, dtype=np.uint32)) & 1).mean(0)
        assert np.abs(bits - 0.5).max() < tol, (seed, bits)
        # the XOR of two steps' words is itself balanced (round 2's generator related them through one constant)
        xb = (((w0 ^ w1)[:, None] >> np.arange(32, dtype=np.uint32)) & 1).mean(0)
        assert np.abs(xb - 0.5).max() < tol
        hi = gat_drop_word(idx[:4096] + np.uint64(1 << 32), 7, seed)
        assert (hi != w0[:4096]).mean() > 0.99
    # a different seed or offset changes about half of the keep decisions
    a, b = gat_drop_word(idx, 0, 5) >= np.uint32(1 << 31), gat_drop_word(idx, 0, 6) >= np.uint32(1 << 31)
    assert abs((a != b).mean() - 0.5) < tol


def check_gat_dropout(eng, dev, oracle):
    """Attention dropout inside the fused op (gat_conv.py:104; GATConvFuse's dropout_rate): the mask is
    rebuilt on the host from the RNG state the launch read, and forward + gradients are compared with the
    in-tree math written in torch with that mask; plus keep-rate, fresh mask per call, eval mode = no-op."""
    rng = np.random.default_rng(31)
    old = eng.chunk
    try:
        for chunk, (N, E, H, C), pd in ((0, (40, 500, 4, 8), 0.4), (8, (30, 400, 8, 4), 0.25), (0, (25, 300, 3, 5), 0.6)):
            eng.chunk = chunk
            eng.graph_cache.clear(); eng.seg_cache.clear()
            index = _rand_graph(rng, N, E)
            index[1, : E // 4] = 2  # a hub row (chunked when chunk = 8)
            it = to_t(index, dev)
            el, er = (to_t(rng.standard_normal((N, H)).astype(np.float32), dev).requires_grad_(True) for _ in range(2))
            x = to_t(rng.standard_normal((N, H, C)).astype(np.float32), dev).requires_grad_(True)
            go = to_t(rng.standard_normal((N, H, C)).astype(np.float32), dev)
            seed, offset = (int(v) for v in eng._rng_state(dev).cpu())
            y = eng.gat_fused(it, el, er, x, 0.2, dropout_rate=pd)
            assert int(eng._rng_state(dev)[1]) == offset + 1
            y.backward(go)
            # host mask per (sorted position, head) -> per original edge through the plan's permutation
            gp = eng.graph_plan(it, N)
            pos = np.arange(E, dtype=np.int64)
            # the word of counter p * H + h  (gat.hip: drop_word)
            draw = gat_drop_word((pos[:, None] * H + np.arange(H)[None, :]).reshape(-1), offset, seed).reshape(E, H)
            keep_pos = draw >= np.uint32(int(pd * 4294967296.0))
            perm = gp.fwd.perm.cpu().numpy().astype(np.int64) if gp.fwd.perm is not None else pos
            keep = np.empty_like(keep_pos)
            keep[perm] = keep_pos
            mask = to_t(keep.astype(np.float32) / np.float32(1.0 - pd), dev)
            # in-tree math in torch (gat_conv.py:103-112 + softmax.py:29-35) with that mask
            el2, er2, x2 = (t.detach().clone().requires_grad_(True) for t in (el, er, x))
            src, dst = it[0], it[1]
            e = torch.nn.functional.leaky_relu(el2[src] + er2[dst], 0.2)
            mx = torch.full((N, H), -3.4028234663852886e38, device=dev).scatter_reduce(
                0, dst.view(-1, 1).expand(E, H), e, "amax", include_self=True)
            ex = torch.exp(e - mx[dst])
            den = torch.zeros(N, H, device=dev).index_add_(0, dst, ex)
            alpha = ex / (den[dst] + 1e-16) * mask
            ref = torch.zeros(N, H, C, device=dev).index_add_(0, dst, alpha.unsqueeze(-1) * x2[src])
            ref.backward(go)
            torch.testing.assert_close(y.detach(), ref.detach(), rtol=2e-5, atol=2e-6)
            torch.testing.assert_close(x.grad, x2.grad, rtol=2e-4, atol=2e-5)
            torch.testing.assert_close(el.grad, el2.grad, rtol=2e-4, atol=5e-5)
            torch.testing.assert_close(er.grad, er2.grad, rtol=2e-4, atol=5e-5)
            assert abs(float(keep_pos.mean()) - (1 - pd)) < 0.08
            y2 = eng.gat_fused(it, el.detach(), er.detach(), x.detach(), 0.2, dropout_rate=pd)
            assert not torch.equal(y2, y.detach()), "every call must draw a new mask"
            y3 = eng.gat_fused(it, el.detach(), er.detach(), x.detach(), 0.2, dropout_rate=pd, training=False)
            np.testing.assert_allclose(to_np(y3), oracle.gat_fwd(index, to_np(el.detach()), to_np(er.detach()),
                                                                to_np(x.detach()), 0.2), rtol=2e-5, atol=2e-6)
    finally:
        eng.chunk = old
        eng.graph_cache.clear(); eng.seg_cache.clear()


def check_gat_random(eng, dev, oracle):
    rng = np.random.default_rng(9)
    for (N, E, H, C) in ((30, 200, 8, 8), (64, 700, 4, 16), (17, 90, 1, 5), (40, 300, 3, 7), (25, 250, 8, 64),
                         # wide heads (C > 16): group-of-lanes backward, every group width, C % 4 != 0 too
                         (20, 150, 2, 41), (20, 150, 2, 24), (18, 140, 3, 40), (30, 260, 8, 40), (16, 120, 1, 100),
                         (12, 100, 1, 300), (22, 200, 16, 20), (15, 130, 4, 68)):
        index = _rand_graph(rng, N, E)
        index[1, :5] = N - 1
        _check_gat(eng, dev, oracle, index, N, H, C, rng)
        _check_gat_separate_buffers(eng, dev, index, N, H, C, rng)
    # isolated destination rows: out = 0, no NaN (den = 0 + 1e-16)
    index = np.array([[0, 1, 2], [1, 1, 1]], dtype=np.int64)
    x = rng.standard_normal((4, 2, 4)).astype(np.float32)
    el = rng.standard_normal((4, 2)).astype(np.float32)
    y = eng.gat_fused(to_t(index, dev), to_t(el, dev), to_t(el, dev), to_t(x, dev), 0.2)
    ynp = to_np(y)
    assert np.isfinite(ynp).all() and (ynp[[0, 2, 3]] == 0).all()
    np.testing.assert_allclose(ynp, oracle.gat_fwd(index, el, el, x, 0.2), rtol=1e-5, atol=1e-6)


def check_edge_cases(eng, dev, oracle):
    f32 = torch.float32
    # empty inputs
    x0 = torch.zeros((0, 4), dtype=f32, device=dev)
    i0 = torch.zeros((0,), dtype=torch.int64, device=dev)
    assert to_np(eng.c_segment_sum(x0, i0, 3)).tolist() == [[0.0] * 4] * 3
    assert to_np(eng.c_segment_mean(x0, i0, 3)).tolist() == [[0.0] * 4] * 3
    mx, arg = eng.segment_max_with_arg(x0, i0, 3)
    assert (to_np(mx) == 0).all() and (to_np(arg) == 0).all()  # segment_max_cpu.cpp:28-30; fill = E = 0
    # N = 0
    assert eng.c_segment_sum(x0, i0, 0).shape == (0, 4)
    # ragged: single element, N > E with empty tail rows, 1-D and 3-D x
    x1 = torch.tensor([2.5], device=dev)
    i1 = torch.tensor([3], device=dev)
    assert to_np(eng.c_segment_sum(x1, i1, 6)).tolist() == [0, 0, 0, 2.5, 0, 0]
    low = float(np.finfo(np.float32).min)
    assert to_np(eng.c_segment_max(x1, i1, 6)).tolist() == [low, low, low, 2.5, low, low]
    x3 = torch.arange(24, dtype=f32, device=dev).reshape(3, 2, 4)
    i3 = torch.tensor([1, 0, 1], device=dev)
    assert_same(to_np(eng.c_segment_mean(x3, i3, 2)), oracle.segment_mean(to_np(x3), to_np(i3), 2), "3-D mean")
    # mean with ids >= E (outside the reference's defined domain): mathematically correct mean
    xm = torch.tensor([[2.0], [4.0], [1.0], [3.0]], device=dev)
    im = torch.tensor([0, 0, 5, 5], device=dev)
    assert to_np(eng.c_segment_mean(xm, im, 6)).reshape(-1).tolist() == [3.0, 0, 0, 0, 0, 2.0]
    # non-contiguous x
    xn = torch.arange(40, dtype=f32, device=dev).reshape(5, 8)[:, ::2]
    i_n = torch.tensor([0, 1, 0, 1, 2], device=dev)
    assert_same(to_np(eng.c_segment_sum(xn, i_n, 3)), oracle.segment_sum(to_np(xn.contiguous()), to_np(i_n), 3), "strided x")
    # errors: same exception types as the reference's TORCH_CHECK_INDEX / dtype checks
    import pytest

    xe = torch.ones((3, 2), device=dev)
    with pytest.raises(IndexError):
        eng.c_segment_max(xe, torch.tensor([0, 5, 1], device=dev), 3)
    with pytest.raises(IndexError):
        eng.c_segment_sum(xe, torch.tensor([0, -1, 1], device=dev), 3)
    with pytest.raises(IndexError):
        eng.c_segment_sum(xe, torch.tensor([[0, 1, 1]], device=dev), 3)  # index.dim() != 1
    with pytest.raises(IndexError):
        eng.c_segment_sum(xe, torch.tensor([0, 1], device=dev), 3)  # size mismatch
    with pytest.raises(RuntimeError, match="Long"):
        eng.c_segment_sum(xe, torch.tensor([0, 1, 1], device=dev, dtype=torch.int32), 3)
    with pytest.raises(RuntimeError, match="Float"):
        eng.c_spmm_sum(torch.tensor([[0, 1], [1, 0]], device=dev), torch.ones(2, device=dev),
                       torch.ones((2, 2), dtype=torch.float64, device=dev))
    with pytest.raises(IndexError):
        eng.c_spmm_sum(torch.tensor([[0, 7], [1, 0]], device=dev), torch.ones(2, device=dev), torch.ones((2, 2), device=dev))


def check_convert(eng, dev, golden=None):
    """ind2ptr / ptr2ind / sort_edge_index vs the reference's numpy statements
    (ops/sparse/__init__.py:23-41: bincount + cumsum, repeat(arange, diff); sort_edge_index.py:36-39:
    argsort(row * N + col)) and, with `golden`, vs what the reference's compiled c_ind2ptr / c_ptr2ind returned
    (tests/golden/convert.npz, ops/sparse/cpu/convert.cpp built by oracle/Makefile)."""
    from gammagl_amd import sparse

    if golden is not None:
        g = golden["convert"]
        for ci in range(int(g["ncases"])):
            ind, M, ptr = g[f"v{ci}_ind"], int(g[f"v{ci}_M"]), g[f"v{ci}_ptr"]
            got = sparse.ind2ptr(to_t(ind, dev), M, eng=eng)
            assert_same(to_np(got), ptr, f"ind2ptr vs reference, case {ci}")
            assert_same(to_np(sparse.ptr2ind(to_t(ptr, dev), len(ind), eng=eng)), ind, f"ptr2ind vs reference, case {ci}")

    rng = np.random.default_rng(3)
    for (M, E) in ((1, 0), (5, 1), (40, 300), (1000, 20000), (7, 5000)):
        ind = np.sort(rng.integers(0, M, size=E)).astype(np.int64)
        ptr_ref = np.concatenate(([0], np.cumsum(np.bincount(ind, minlength=M), dtype=np.int64)))
        ptr = sparse.ind2ptr(to_t(ind, dev), M, eng=eng)
        assert_same(to_np(ptr), ptr_ref, f"ind2ptr M{M} E{E}")
        uns = rng.permutation(ind)  # the numpy fallback (bincount) accepts unsorted ind too
        assert_same(to_np(sparse.ind2ptr(to_t(uns, dev), M, eng=eng)), ptr_ref, "ind2ptr unsorted")
        ind_ref = np.repeat(np.arange(M, dtype=np.int64), np.diff(ptr_ref))
        assert_same(to_np(sparse.ptr2ind(ptr, E, eng=eng)), ind_ref[:E], f"ptr2ind M{M} E{E}")
        assert_same(to_np(sparse.ptr2ind(ptr, None, eng=eng)), ind_ref, "ptr2ind E=None")
    import pytest

    with pytest.raises(IndexError):
        sparse.ind2ptr(to_t(np.array([0, 9]), dev), 5, eng=eng)
    for (N, E) in ((6, 0), (10, 50), (300, 5000)):
        ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
        attr = rng.standard_normal((E, 3)).astype(np.float32)
        for by_row in (True, False):
            key = ei[1 - int(by_row)] * N + ei[int(by_row)]
            perm = np.argsort(key, kind="stable")
            out, a = sparse.sort_edge_index(to_t(ei, dev), to_t(attr, dev), N, by_row, eng=eng)
            assert_same(to_np(out), ei[:, perm], f"sort_edge_index N{N} by_row={by_row}")
            assert_same(to_np(a), attr[perm], "sort_edge_index attr")
        out2 = sparse.sort_edge_index(to_t(ei, dev), eng=eng) if E > 0 else None  # num_nodes inferred
        if out2 is not None:
            n2 = int(ei.max()) + 1
            assert_same(to_np(out2), ei[:, np.argsort(ei[0] * n2 + ei[1], kind="stable")], "inferred N")
    # FusedGATConv's own preprocessing (fusedgat_conv.py:103-117) reproduced on the device
    N, E = 50, 400
    ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
    s = sparse.sort_edge_index(to_t(ei, dev), num_nodes=N, eng=eng)
    row_ptr = sparse.ind2ptr(s[0], N, eng=eng)
    assert int(row_ptr[-1]) == E and bool((row_ptr[1:] >= row_ptr[:-1]).all())
    s2, permute = sparse.sort_edge_index(s, torch.arange(E, device=dev), N, sort_by_row=False, eng=eng)
    col_ptr = sparse.ind2ptr(s2[1], N, eng=eng)
    assert torch.equal(s[:, permute], s2) and int(col_ptr[-1]) == E


def check_sampler(eng, dev, oracle):
    """sample_adj / NeighborSampler: the deterministic branch bit-for-bit vs the restated reference
    (sample.cpp:39-55,104-130); the random branches through the invariants the reference guarantees
    (distinct positions, min(deg, fanout) per row, columns sorted by local id, seeds first) and a
    uniformity check of Floyd's algorithm on Philox."""
    import ctypes

    from gammagl_amd import sampler
    from gammagl_amd.ops import _ptr

    rng = np.random.default_rng(12)
    N, E = 300, 4000
    ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
    ei[1, :60] = 7  # one heavy row
    order = np.argsort(ei[1], kind="stable")
    rowptr = np.concatenate(([0], np.cumsum(np.bincount(ei[1], minlength=N)))).astype(np.int64)
    col = ei[0][order]
    rp, cl = to_t(rowptr, dev), to_t(col, dev)
    perm_nodes = rng.permutation(N)
    seeds = np.concatenate(([7], perm_nodes[perm_nodes != 7][:63])).astype(np.int64)  # unique, heavy row first
    idx = to_t(seeds, dev)
    # (1) no sampling: exact
    got = sampler.sample_adj(rp, cl, idx, -1, eng=eng)
    ref = oracle.sample_adj_full(rowptr, col, seeds)
    for a, b, nm in zip(got, ref, ("rowptr", "col", "n_id", "e_id")):
        assert_same(to_np(a), b, "sample_adj full " + nm)
    # duplicate seeds stay verbatim in n_id and neighbours map to the LAST copy (sample.cpp:24-29), both with the
    # relabel scratch and without it
    dup = np.array([7, 3, 9, 3, 100, 7], np.int64)
    refd = oracle.sample_adj_full(rowptr, col, dup)
    for fpos in (None, torch.full((N,), sampler._BIG, dtype=torch.int64, device=dev)):
        gotd = sampler.sample_adj(rp, cl, to_t(dup, dev), -1, eng=eng, first_pos=fpos)
        for a, b, nm in zip(gotd, refd, ("rowptr", "col", "n_id", "e_id")):
            assert_same(to_np(a), b, "sample_adj duplicate seeds " + nm)
        if fpos is not None:
            assert bool((fpos == sampler._BIG).all())
    deg = rowptr[seeds + 1] - rowptr[seeds]
    for fanout, replace in ((5, False), (25, False), (10, True)):
        orp, ocol, n_id, e_pos = (to_np(t) for t in sampler.sample_adj(rp, cl, idx, fanout, replace, eng=eng))
        k = np.diff(orp)
        want = np.where(deg > 0, fanout, 0) if replace else np.minimum(deg, fanout)
        assert (k == want).all()
        assert (n_id[: len(seeds)] == seeds).all() and len(np.unique(n_id)) == len(n_id)
        assert (n_id[ocol] == col[e_pos]).all()  # local ids map back to the sampled neighbours
        for i, s in enumerate(seeds):
            pos = e_pos[orp[i]:orp[i + 1]]
            assert ((pos >= rowptr[s]) & (pos < rowptr[s + 1])).all()
            if not replace:
                assert len(np.unique(pos)) == len(pos)
                if deg[i] <= fanout:
                    assert sorted(pos.tolist()) == list(range(rowptr[s], rowptr[s + 1]))
            assert (np.diff(ocol[orp[i]:orp[i + 1]]) >= 0).all()  # sorted by local id
    # (2) uniformity of the without-replacement draw on the heavy row (deg 60+, fanout 10)
    B, f = 4000, 10
    sd = torch.full((B,), 7, dtype=torch.int64, device=dev)
    d7 = int(rowptr[8] - rowptr[7])
    orp = torch.arange(0, (B + 1) * f, f, dtype=torch.int64, device=dev)
    e_pos = torch.empty(B * f, dtype=torch.int64, device=dev)
    nbr = torch.empty(B * f, dtype=torch.int64, device=dev)
    eng._check(eng.lib.ggl_sample_pick(_ptr(rp), _ptr(cl), _ptr(sd), B, f, 0, _ptr(orp), _ptr(eng._rng_state(dev)),
                                       _ptr(e_pos), _ptr(nbr), eng._stream(dev)))
    cnt = np.bincount(to_np(e_pos) - rowptr[7], minlength=d7)
    exp = B * f / d7
    assert cnt.sum() == B * f and np.abs(cnt - exp).max() < 6 * np.sqrt(exp), (cnt.min(), cnt.max(), exp)
    assert (to_np(e_pos).reshape(B, f)[0] != to_np(e_pos).reshape(B, f)[1]).any()  # rows draw independently
    # (3) two-hop NeighborSampler + the aggregate straight from the block's CSR (no sort, no plan sync)
    ns = sampler.NeighborSampler(to_t(ei, dev), [5, 3], num_nodes=N, eng=eng)
    batch, n_id, adjs = ns.sample(seeds[:16])
    assert len(adjs) == 2 and adjs[1].size[1] == 16 and adjs[0].size[1] == adjs[1].size[0]
    assert adjs[0].size[0] == n_id.shape[0]
    for adj in adjs:
        src_l, dst_l = adj.edge_index
        assert int(src_l.max()) < adj.size[0] and int(dst_l.max()) < adj.size[1]
        assert bool((to_t(ei, dev)[:, adj.e_id][1] >= 0).all())
    adj = adjs[0]
    x = torch.randn(adj.size[0], 12, generator=torch.Generator().manual_seed(0)).to(dev)
    plan = eng.plan_from_rowptr(adj.rowptr, adj.edge_index.shape[1], max_len=adj.fanout)
    got = eng.segment_reduce(x[adj.edge_index[0]], plan, "mean")
    ref = eng.c_segment_mean(x[adj.edge_index[0]], adj.edge_index[1].contiguous(), adj.size[1])
    assert torch.equal(got, ref)
    # the global edges the block refers to are real edges between the right nodes
    g_src, g_dst = to_t(ei, dev)[0][adj.e_id], to_t(ei, dev)[1][adj.e_id]
    assert torch.equal(n_id[adj.edge_index[0]], g_src)


def block_triples(orp, ocol, nid, eid):
    """(seed row, global neighbour, e_id) of every edge of a sampled block, sorted: what the block MEANS, whatever the
    numbering of its new nodes and the order of equal columns (std::sort is not stable, sample.cpp:112-118)."""
    orp, ocol, nid, eid = (np.asarray(a, np.int64) for a in (orp, ocol, nid, eid))
    rows = np.repeat(np.arange(len(orp) - 1, dtype=np.int64), np.diff(orp))
    t = np.stack([rows, nid[ocol] if len(ocol) else ocol, eid], 1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def ties_sorted(orp, ocol, eid):
    """e_id with the pairs of every row ordered by (local column, e_id) — the reference leaves ties in std::sort's order"""
    orp, ocol, eid = (np.asarray(a, np.int64) for a in (orp, ocol, eid))
    rows = np.repeat(np.arange(len(orp) - 1, dtype=np.int64), np.diff(orp))
    return eid[np.lexsort((eid, ocol, rows))]


def compare_block_with_reference(got, g, k, what):
    """`got` = (rowptr, col, n_id, e_id) of a sample_adj implementation, g[k_*] = what the reference's own
    c_sample_adj returned for the same input (tests/golden/sampler.npz)."""
    orp, ocol, nid, eid = (np.asarray(a, np.int64) for a in got)
    seeds, fanout = g[k + "_seeds"], int(g[k + "_fanout"])
    r_orp, r_col, r_nid, r_eid = g[k + "_orp"], g[k + "_ocol"], g[k + "_nid"], g[k + "_eid"]
    assert_same(orp, r_orp, what + " rowptr")
    if fanout < 0:   # no sampling: everything is determined (ties in e_id up to std::sort's order)
        assert_same(ocol, r_col, what + " col")
        assert_same(nid, r_nid, what + " n_id")
        assert_same(ties_sorted(orp, ocol, eid), ties_sorted(r_orp, r_col, r_eid), what + " e_id")
        return
    # every neighbour taken through the sampled code path: new nodes numbered in unordered_set order
    assert_same(nid[: len(seeds)], seeds, what + " seeds first")
    assert_same(np.sort(nid), np.sort(r_nid), what + " node set")
    assert_same(block_triples(orp, ocol, nid, eid), block_triples(r_orp, r_col, r_nid, r_eid), what + " edges")
    rows = np.repeat(np.arange(len(orp) - 1), np.diff(orp))
    if len(ocol) > 1:
        inner = rows[1:] == rows[:-1]
        assert (np.diff(ocol)[inner] >= 0).all(), what + " columns ascending"


def check_sampler_golden(eng, dev, golden):
    """sample_adj against the outputs of the reference's own c_sample_adj (built from ops/sparse/cpu/sample.cpp by
    oracle/Makefile, run by tests/golden/make_golden.py): the two branches without a random draw."""
    from gammagl_amd import sampler

    g = golden["sampler"]
    for ci in range(int(g["ncases"])):
        k = f"c{ci}"
        rp, cl, idx = to_t(g[k + "_rowptr"], dev), to_t(g[k + "_col"], dev), to_t(g[k + "_seeds"], dev)
        got = sampler.sample_adj(rp, cl, idx, int(g[k + "_fanout"]), False, eng=eng)
        compare_block_with_reference([to_np(t) for t in got], g, k, f"sampler case {ci}")


def check_colsum(eng, dev):
    """bias-gradient kernel: column sums vs an f64 sum; also through BiasAdd's autograd."""
    g = torch.Generator(device="cpu").manual_seed(2)
    for (N, K) in ((0, 5), (1, 1), (7, 3), (1000, 47), (5000, 256), (3000, 300), (70000, 16), (300000, 4)):  # the last two reduce their partials recursively
        x = torch.randn(N, K, generator=g).to(dev)
        got = eng.colsum(x)
        ref = x.double().sum(0)
        bound = x.double().abs().sum(0)
        assert got.shape == (K,)
        assert bool(((got.double() - ref).abs() <= 1e-6 * bound + 1e-30).all()), (N, K)
        assert torch.equal(got, eng.colsum(x))  # deterministic
    x = torch.randn(300, 47, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(1, 47, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(300, 47, generator=g).to(dev)
    y = eng.bias_add(x, b)
    y.backward(go)
    assert torch.equal(y.detach(), x.detach() + b.detach()) and torch.equal(x.grad, go)
    torch.testing.assert_close(b.grad, go.sum(0, keepdim=True), rtol=1e-5, atol=1e-5)


def check_bias_act(eng, dev):
    """fused bias + ReLU + dropout: exact vs torch without dropout; with dropout the mask statistics,
    the 1/(1-p) scaling, mask consistency between forward and backward, and a fresh mask per call."""
    g = torch.Generator(device="cpu").manual_seed(5)
    for (N, K) in ((0, 4), (1, 1), (33, 47), (500, 256), (257, 6)):
        a = torch.randn(N, K, generator=g).to(dev).requires_grad_(True)
        b = torch.randn(1, K, generator=g).to(dev).requires_grad_(True)
        go = torch.randn(N, K, generator=g).to(dev)
        for relu in (False, True):
            a.grad = b.grad = None
            y = eng.bias_act(a, b, relu=relu, p_drop=0.0)
            ref = a.detach() + b.detach()
            ref = torch.relu(ref) if relu else ref
            assert torch.equal(y.detach(), ref), (N, K, relu)
            if N > 0:
                y.backward(go)
                gref = go * (ref > 0) if relu else go
                assert torch.equal(a.grad, gref)
                torch.testing.assert_close(b.grad, gref.sum(0, keepdim=True), rtol=1e-5, atol=1e-5)
    y0 = eng.bias_act(torch.randn(7, 3, generator=g).to(dev), None, relu=True)  # bias=None
    assert bool((y0 >= 0).all())
    N, K, p = 4000, 64, 0.5
    a = (torch.rand(N, K, generator=g) + 0.5).to(dev).requires_grad_(True)  # strictly positive: relu inactive
    y = eng.bias_act(a, None, relu=True, p_drop=p, training=True)
    kept = y.detach() != 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.01, frac
    torch.testing.assert_close(y.detach()[kept], (a.detach() / (1 - p))[kept])
    col_frac = kept.float().mean(0)
    assert float((col_frac - (1 - p)).abs().max()) < 0.05  # no column- or row-structured mask
    y.backward(torch.ones_like(y))
    assert torch.equal(a.grad != 0, kept) and torch.allclose(a.grad[kept], torch.tensor(1 / (1 - p), device=dev))
    y2 = eng.bias_act(a.detach(), None, relu=True, p_drop=p, training=True)
    assert not torch.equal(y2 != 0, kept), "every call must draw a new mask"
    assert torch.equal(eng.bias_act(a.detach(), None, relu=True, p_drop=p, training=False), a.detach())


def check_spmm_bias_act(eng, dev):
    """SpMM with the layer epilogue applied in its store (ggl_spmm_sum_bias_act) == SpMM, then the epilogue
    kernel, replayed on the same RNG state: values, dropout mask and every gradient bit for bit — short
    rows, chunked hub rows (epilogue in long_final_kernel), wave-per-row and narrow widths, K % 4 != 0."""
    g = torch.Generator(device="cpu").manual_seed(21)
    old = eng.chunk
    try:
        for chunk in (0, 8):
            eng.chunk = chunk
            eng.graph_cache.clear(); eng.seg_cache.clear()
            for (N, E, K) in ((40, 600, 8), (64, 900, 64), (50, 700, 256), (30, 300, 47), (5, 0, 4)):
                ei = torch.randint(0, N, (2, E), generator=g)
                if E:
                    ei[1, : E // 3] = 3  # a hub row (chunked when chunk = 8)
                ei = ei.to(dev)
                w = torch.rand(E, generator=g).to(dev)
                gp = eng.graph_plan(ei, N)
                go = torch.randn(N, K, generator=g).to(dev)
                for (relu, p) in ((False, 0.0), (True, 0.0), (True, 0.5), (False, 0.3)):
                    xa = torch.randn(N, K, generator=g).to(dev).requires_grad_(True)
                    ba = torch.randn(1, K, generator=g).to(dev).requires_grad_(True)
                    xb, bb = xa.detach().clone().requires_grad_(True), ba.detach().clone().requires_grad_(True)
                    st = eng._rng_state(dev).clone()
                    ya = eng.spmm_bias_act(gp, w, xa, ba, relu=relu, p_drop=p, training=True)
                    eng._rng_state(dev).copy_(st)   # replay the same mask for the two-kernel form
                    yb = eng.bias_act(eng.spmm(gp, w, xb), bb, relu=relu, p_drop=p, training=True)
                    assert torch.equal(ya, yb), (chunk, N, E, K, relu, p)
                    if N * K:
                        ya.backward(go)
                        yb.backward(go)
                        assert torch.equal(xa.grad, xb.grad) and torch.equal(ba.grad, bb.grad)
                    if p > 0 and E:
                        assert bool((ya == 0).any()) and bool((ya != 0).any())
        # no bias, no weights
        eng.chunk = old
        ei = torch.randint(0, 20, (2, 100), generator=g).to(dev)
        gp = eng.graph_plan(ei, 20)
        x = torch.randn(20, 12, generator=g).to(dev)
        assert torch.equal(eng.spmm_bias_act(gp, None, x, None, relu=True), torch.relu(eng.spmm(gp, None, x)))
        assert torch.equal(eng.spmm_bias_act(gp, None, x, None, relu=True, p_drop=0.9, training=False),
                           torch.relu(eng.spmm(gp, None, x)))
    finally:
        eng.chunk = old
        eng.graph_cache.clear(); eng.seg_cache.clear()


def check_strided_accumulate(eng, dev, oracle):
    """ggl_spmm_sum_ex / ggl_segment_sum_ex: column blocks of wider matrices read and written in place
    (row strides), and out += (accumulate) — against the dense ops on contiguous copies."""
    g = torch.Generator(device="cpu").manual_seed(17)
    old = eng.chunk
    try:
        for chunk in (0, 8):
            eng.chunk = chunk
            eng.graph_cache.clear(); eng.seg_cache.clear()
            N, M, E, KW = 37, 29, 500, 24
            ei = torch.stack([torch.randint(0, M, (E,), generator=g), torch.randint(0, N, (E,), generator=g)])
            ei[1, :150] = 5
            ei = ei.to(dev)
            w = torch.rand(E, generator=g).to(dev)
            gp = eng.graph_plan(ei, N, M)
            xw = torch.randn(M, KW, generator=g).to(dev)
            for (c0, c1) in ((0, 8), (8, 24), (4, 16), (0, 24)):
                xs = xw[:, c0:c1]
                dense, _ = eng._spmm_fwd("sum", gp.fwd, gp.col, w, xs.contiguous(), N)
                outw = torch.full((N, KW), 7.0, device=dev)
                eng.spmm_sum_into(gp.fwd, gp.col, w, xs, outw[:, c0:c1])
                assert torch.equal(outw[:, c0:c1], dense)
                assert bool((outw[:, :c0] == 7).all()) and bool((outw[:, c1:] == 7).all())  # nothing else touched
                base = torch.randn(N, KW, generator=g).to(dev)
                acc = base.clone()
                eng.spmm_sum_into(gp.fwd, gp.col, w, xs, acc[:, c0:c1], accumulate=True)
                torch.testing.assert_close(acc[:, c0:c1], base[:, c0:c1] + dense, rtol=1e-5, atol=1e-5)
                assert torch.equal(acc[:, :c0], base[:, :c0]) and torch.equal(acc[:, c1:], base[:, c1:])
            # ragged widths (47 columns out of a 50-wide matrix, rows that are not 16-byte pieces): written in place,
            # and accumulated — the ragged kernels' tail lane shares columns with its neighbour, which an accumulate
            # must not see twice (it takes the one-element-per-lane kernels)
            xr = torch.randn(M, 50, generator=g).to(dev)
            dense, _ = eng._spmm_fwd("sum", gp.fwd, gp.col, w, xr[:, 1:48].contiguous(), N)
            outr = torch.full((N, 50), 7.0, device=dev)
            eng.spmm_sum_into(gp.fwd, gp.col, w, xr[:, 1:48], outr[:, 2:49])
            assert torch.equal(outr[:, 2:49], dense) and bool((outr[:, :2] == 7).all()) and bool((outr[:, 49:] == 7).all())
            base = torch.randn(N, 50, generator=g).to(dev)
            accr = base.clone()
            eng.spmm_sum_into(gp.fwd, gp.col, w, xr[:, 1:48], accr[:, 2:49], accumulate=True)
            torch.testing.assert_close(accr[:, 2:49], base[:, 2:49] + dense, rtol=1e-5, atol=1e-5)
            assert torch.equal(accr[:, :2], base[:, :2]) and torch.equal(accr[:, 49:], base[:, 49:])
            # segment_sum over a strided message block, accumulated onto a column block
            ids = ei[1].contiguous()
            plan = eng.seg_plan(ids, N)
            mw = torch.randn(E, KW, generator=g).to(dev)
            base = torch.randn(N, KW, generator=g).to(dev)
            acc = base.clone()
            eng.segment_sum_into(mw[:, 4:12], plan, acc[:, 8:16], accumulate=True)
            ref = oracle.segment_sum(to_np(mw[:, 4:12].contiguous()), to_np(ids), N)
            np.testing.assert_allclose(to_np(acc[:, 8:16]), to_np(base[:, 8:16]) + ref, rtol=1e-5, atol=1e-5)
            assert torch.equal(acc[:, :8], base[:, :8]) and torch.equal(acc[:, 16:], base[:, 16:])
    finally:
        eng.chunk = old
        eng.graph_cache.clear(); eng.seg_cache.clear()


def check_bspmm_wide(eng, dev):
    """bspmm forward, input gradient and WEIGHT gradient for wide / odd channel counts (the weight gradient has a
    lanes-split-channels kernel for C > 16 on the GPU) against the formula written out in torch."""
    g = torch.Generator(device="cpu").manual_seed(13)
    for (N, E, H, C) in ((30, 400, 2, 24), (25, 300, 1, 100), (40, 500, 8, 44), (20, 260, 3, 20), (18, 200, 8, 41),
                         (16, 150, 1, 300), (22, 240, 4, 8)):
        ei = torch.randint(0, N, (2, E), generator=g).to(dev)
        x = torch.randn(N, H, C, generator=g).to(dev).requires_grad_(True)
        w = torch.rand(E, H, generator=g).to(dev).requires_grad_(True)
        go = torch.randn(N, H, C, generator=g).to(dev)
        y = eng.c_bspmm_sum(ei, w, x)
        y.backward(go)
        xr, wr = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        ref = torch.zeros(N, H, C, device=dev).index_add_(0, ei[1], xr[ei[0]] * wr.unsqueeze(-1))
        ref.backward(go)
        torch.testing.assert_close(y.detach(), ref.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(x.grad, xr.grad, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(w.grad, wr.grad, rtol=1e-5, atol=1e-5)


def check_schedule_invariance(eng, dev):
    """Scheduling knobs never change a bit: the row hand-out order (global sort by length, id windows, none) and the XCD
    run mapping (off, runs of 64 / 2048 rows, one contiguous eighth) give identical sums, means, maxima, argmax witnesses
    and fused epilogues on a graph with a hub, empty rows and more row blocks than 16 runs."""
    g = torch.Generator().manual_seed(17)
    N, E = 6000, 90000
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 50, (E,), generator=g)
    dst[:3000] = 11
    ei = torch.stack([src, dst]).to(dev)
    w = torch.rand(E, generator=g).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    xs = {K: torch.randn(N, K, generator=g).to(dev) for K in (64, 16, 256)}
    msg = torch.randn(E, 24, generator=g).to(dev)
    old = (eng.row_order_window, eng.xcd_run_rows, eng.lib.ggl_get_option(b"xcd_swizzle"), eng.lib.ggl_get_option(b"row_order"))
    ref = None
    try:
        for win, run, swz, ro in ((0, 0, 0, 1), (512, 0, 0, 1), (2048, 64, 0, 1), (2048, 2048, 0, 1), (512, 0, 1, 1), (512, 0, 8, 1),
                                  (512, 64, 0, 0), (2048, -1, 0, 1)):
            eng.clear_caches()
            eng.row_order_window, eng.xcd_run_rows = win, run
            eng.set_option("xcd_swizzle", swz)
            eng.set_option("row_order", ro)
            gp = eng.graph_plan(ei, N)
            assert gp.fwd.xcd_run == (run if run >= 0 else 0)        # (auto: 90 000 edges is below the size where it decides)
            got = []
            for K, x in xs.items():
                out = torch.empty(N, K, device=dev)
                eng.spmm_sum_into(gp.fwd, gp.col, w, x, out)
                got.append(out.clone())
                eng.spmm_sum_into(gp.bwd, gp.colT, w, x, out)
                got.append(out.clone())
            y = torch.empty(N, 64, device=dev)
            eng.spmm_epi_into(gp.fwd, gp.col, w, xs[64], y, mean=True, bias=bias, relu=True)
            got.append(y)
            got.append(eng.c_spmm_max(ei, w, xs[16]))
            mx, arg = eng.segment_max_with_arg(msg, ei[1].contiguous(), N)
            got += [mx, arg, eng.c_segment_sum(msg, ei[1].contiguous(), N), eng.c_segment_mean(msg, ei[1].contiguous(), N)]
            if ref is None:
                ref = got
            else:
                for i, (a, b) in enumerate(zip(ref, got)):
                    assert torch.equal(a, b), (win, run, swz, ro, i)
    finally:
        eng.row_order_window, eng.xcd_run_rows = old[0], old[1]
        eng.set_option("xcd_swizzle", old[2])
        eng.set_option("row_order", old[3])
        eng.clear_caches()


def check_half_ragged_rows(eng, dev, oracle):
    """f16 / bf16 segment sum / mean on rows that are not made of aligned 16-byte pieces (12, 13, 47, 100 columns, and a
    base address that is only 2-byte aligned): eight elements per lane with a ragged last lane
    (reduce.hip RowIO<uint16_t, 8, true>) — the oracle's serial storage-dtype sums bit for bit, incl. a hub row and
    with the lanes switched off (A/B)."""
    rng = np.random.default_rng(31)
    old = eng.lib.ggl_get_option(b"ragged4")
    try:
        for dt in ("float16", "bfloat16"):
            for K in (12, 13, 47, 100):
                N, E = 70, 4000
                ids = rng.integers(0, N - 4, E).astype(np.int64)
                ids[: E // 4] = 3
                xf = (rng.standard_normal((E + 1, K)) * 3 + 0.25).astype(np.float32)
                xh = oracle.f32_to_bf16_bits(xf) if dt == "bfloat16" else xf.astype(np.float16)
                x_all = to_t(xh, dev, dt)
                x = x_all[1:]                       # contiguous, base = K * 2 bytes past an allocation: 2-byte aligned for odd K
                want_s = oracle.segment_sum(xh[1:], ids, N, bf16=dt == "bfloat16")
                want_m = oracle.segment_mean(xh[1:], ids, N, bf16=dt == "bfloat16")
                it = to_t(ids, dev)
                for rag in (1, 0):
                    eng.set_option("ragged4", rag)
                    assert_same(to_np(eng.c_segment_sum(x, it, N)), want_s, f"{dt} K{K} sum ragged={rag}")
                    assert_same(to_np(eng.c_segment_mean(x, it, N)), want_m, f"{dt} K{K} mean ragged={rag}")
                    mx, arg = eng.segment_max_with_arg(x, it, N)       # (ragged lanes from 72 columns up; 32-bit witnesses)
                    omx, oarg = oracle.segment_max(xh[1:], ids, N, bf16=dt == "bfloat16")
                    assert_same(to_np(mx), omx, f"{dt} K{K} max ragged={rag}")
                    assert np.array_equal(to_np(arg), oarg), f"{dt} K{K} argmax ragged={rag}"
    finally:
        eng.set_option("ragged4", old)


def check_bspmm_gradw_sorted(eng, dev, oracle):
    """bspmm's weight gradient along the destination-sorted plan (edgedot.hip: strips staged through LDS on the GPU;
    the plain walk in the host-emulation build): bit for bit the oracle's serial-over-c sums, for heads of 20 ... 300
    channels, with and without the 64-column-block launches that carry the running dot in a scratch buffer, and equal
    to the thread-per-item kernel it replaces.  Graphs with empty rows, duplicate edges and a hub."""
    import numpy as np

    rng = np.random.default_rng(21)
    names = (b"col_block_min_edges", b"col_block_min_degree")
    old = [eng.lib.ggl_get_option(n) for n in names]
    try:
        for blocks in (False, True):
            if blocks:        # force the column-block launches on these toy graphs
                eng.set_option("col_block_min_edges", 0)
                eng.set_option("col_block_min_degree", 0)
            for (N, E, H, C) in ((50, 3000, 1, 256), (40, 900, 8, 44), (33, 700, 2, 136), (21, 500, 3, 20),
                                 (64, 1500, 1, 300), (30, 257, 4, 32), (9, 1, 1, 64),
                                 (40, 1200, 16, 16), (25, 600, 32, 8), (25, 600, 8, 8)):   # wide rows of narrow heads
                src = rng.integers(0, N, E)
                dst = rng.integers(0, max(N - 5, 1), E)              # the last rows stay empty
                dst[: E // 3] = 2                                      # a hub row
                index = np.stack([src, dst]).astype(np.int64)
                w = rng.standard_normal((E, H)).astype(np.float32)
                x = rng.standard_normal((N, H, C)).astype(np.float32)
                go = rng.standard_normal((N, H, C)).astype(np.float32)
                ogx, ogw = oracle.bspmm_sum_bwd(index, w, x, go)
                got = {}
                for sorted_walk in (True, False):
                    eng.gradw_sorted = sorted_walk
                    wt = to_t(w, dev).requires_grad_(True)
                    xt = to_t(x, dev).requires_grad_(True)
                    eng.c_bspmm_sum(to_t(index, dev), wt, xt).backward(to_t(go, dev))
                    got[sorted_walk] = to_np(wt.grad)
                    assert_same(to_np(xt.grad), ogx, f"bspmm gx H{H} C{C}")
                assert_same(got[True], ogw, f"bspmm gw (sorted walk, blocks={blocks}) H{H} C{C}")
                assert_same(got[False], ogw, f"bspmm gw (edge order) H{H} C{C}")
    finally:
        eng.gradw_sorted = True
        for n, v in zip(names, old):
            eng.set_option(n.decode(), v)
    # a plan built from CSR + CSC (no edge_index): rowidx comes from the row pointer
    N, H, C = 12, 2, 24
    g = torch.Generator().manual_seed(3)
    ei = torch.randint(0, N, (2, 90), generator=g)
    o = torch.argsort(ei[1] * N + ei[0], stable=True)
    ei = ei[:, o].contiguous().to(dev)
    gp = eng.graph_plan(ei, N)
    gp2 = eng.graph_plan_from_csr(gp.fwd.rowptr.clone(), gp.col.clone(), gp.bwd.rowptr.clone(), gp.colT.clone(), gp.posT.clone())
    assert torch.equal(gp2.rowidx.long(), ei[1])
    # ... and its weight gradient takes the sorted route for ANY channel count (there is no COO list to walk): narrow
    # and odd heads included, with the switch for the sorted walk off as well (round 3 raised here)
    for C2, sorted_walk in ((8, True), (16, True), (5, True), (8, False), (24, False)):
        eng.gradw_sorted = sorted_walk
        try:
            xs = torch.randn(N, H, C2, generator=g).to(dev)
            ws = torch.rand(90, H, generator=g).to(dev)
            gos = torch.randn(N, H, C2, generator=g).to(dev)
            res = []
            for plan in (gp, gp2):
                wt, xt = ws.clone().requires_grad_(True), xs.clone().requires_grad_(True)
                eng.BSpMMSum.apply(plan, wt, xt).backward(gos)
                res.append((wt.grad, xt.grad))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (C2, sorted_walk)
        finally:
            eng.gradw_sorted = True
    # weighted gspmm over the CSR-built plan: the transposed walk reads the (CSR-ordered) weights through `permute`
    # — twice, so the second call streams the copy sorted into CSC order
    for reduce in ("sum", "mean", "max"):
        xs = torch.randn(N, 6, generator=g).to(dev)
        ws = torch.rand(90, generator=g).to(dev)
        gos = torch.randn(N, 6, generator=g).to(dev)
        for _ in range(2):
            res = []
            for plan in (gp, gp2):
                xt = xs.clone().requires_grad_(True)
                y = eng.spmm(plan, ws, xt, reduce)
                y.backward(gos)
                res.append((y.detach(), xt.grad))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), reduce


def check_plan_cache(eng, dev):
    eng.seg_cache.clear()
    ids = torch.tensor([2, 0, 1, 0, 2, 2], device=dev)
    x = torch.arange(12, dtype=torch.float32, device=dev).reshape(6, 2)
    b0 = eng.stats["plans_built"]
    y1 = eng.c_segment_sum(x, ids, 3)
    y2 = eng.c_segment_sum(x, ids, 3)
    assert eng.stats["plans_built"] == b0 + 1, "second call must hit the plan cache"
    assert torch.equal(y1, y2)
    ids[0] = 1  # in-place edit bumps the version counter -> stale plan must not be reused
    y3 = eng.c_segment_sum(x, ids, 3)
    assert eng.stats["plans_built"] == b0 + 2
    assert to_np(y3).tolist() == [[8.0, 10.0], [4.0, 6.0], [18.0, 20.0]]
    # a view of the same storage (edge_index[1]) hits the same entry
    ei = torch.stack([ids, ids]).contiguous()
    eng.c_segment_sum(x, ei[1], 3)
    b1 = eng.stats["plans_built"]
    eng.c_segment_sum(x, ei[1], 3)
    assert eng.stats["plans_built"] == b1
    # the caches are bounded by the bytes their plans hold, not only by entry count, and can be dropped
    from gammagl_amd.ops import _PlanCache
    small = _PlanCache(cap=16, max_bytes=3000)
    keep = []
    for i in range(6):
        t = torch.arange(100, device=dev) + i
        keep.append(t)
        small.put(t, (i,), torch.zeros(200, dtype=torch.float32, device=dev))   # 800 bytes each
    assert len(small.d) == 3 and small.bytes <= 3000 and small.get(keep[-1], (5,)) is not None
    assert small.get(keep[0], (0,)) is None
    eng.clear_caches()
    assert len(eng.seg_cache.d) == 0 and eng.seg_cache.bytes == 0 and len(eng.graph_cache.d) == 0
    # an edit behind the version counter (.data) is invisible to the key: the opt-in checksum catches it
    v = _PlanCache(cap=4)
    v.verify = True
    t = torch.tensor([3, 1, 2], device=dev)
    v.put(t, (), "plan")
    assert v.get(t, ()) == "plan"
    t.data[0] = 0
    try:
        v.get(t, ())
        raise AssertionError("stale plan returned")
    except RuntimeError as ex:
        assert "clear_caches" in str(ex)


def check_dropout_without_relu_gradient(eng, dev):
    """relu=False with p_drop > 0: a KEPT negative activation must receive its gradient (the first version
    rebuilt the mask as y > 0 and zeroed all of them).  The backward redraws the Philox mask from the rng state
    the forward read, so it is exact also for kept activations that are exactly zero."""
    g = torch.Generator(device="cpu").manual_seed(31)
    for (N, K) in ((200, 64), (131, 47)):
        a0 = torch.randn(N, K, generator=g)
        a0[::7] = 0.0                                    # exact zeros that the mask keeps or drops
        b0 = torch.zeros(1, K)
        go = torch.randn(N, K, generator=g).to(dev)
        a = a0.to(dev).requires_grad_(True)
        b = b0.to(dev).requires_grad_(True)
        p = 0.5
        st = eng._rng_state(dev).clone()
        y = eng.bias_act(a, b, relu=False, p_drop=p)
        y.backward(go)
        # the mask, independently: a second draw from the same state on an input without zeros
        eng._rng_state(dev).copy_(st)
        keep = eng.bias_act(torch.ones(N, K, device=dev), None, relu=False, p_drop=p) != 0
        assert 0.4 < float(keep.float().mean()) < 0.6
        assert torch.equal(y.detach(), torch.where(keep, (a.detach() + b.detach()) * (1 / (1 - p)), torch.zeros_like(y)))
        want = torch.where(keep, go * (1 / (1 - p)), torch.zeros_like(go))
        assert torch.equal(a.grad, want)
        neg_kept = keep & (a.detach() < 0)
        assert int(neg_kept.sum()) > 100 and bool((a.grad[neg_kept] != 0).all())
        zero_kept = keep & (a.detach() == 0)
        assert int(zero_kept.sum()) > 0 and bool((a.grad[zero_kept] != 0).all())
        torch.testing.assert_close(b.grad, want.sum(0, keepdim=True), rtol=1e-5, atol=1e-5)
        # the same through the SpMM-fused epilogue
        E = 900
        ei = torch.randint(0, N, (2, E), generator=g).to(dev)
        gp = eng.graph_plan(ei, N)
        x = torch.randn(N, K, generator=g).to(dev).requires_grad_(True)
        eng._rng_state(dev).copy_(st)
        y2 = eng.spmm_bias_act(gp, None, x, None, relu=False, p_drop=p)
        y2.backward(go)
        gx_ref = eng.spmm(eng.graph_plan(ei.flip(0).contiguous(), N), None, want)
        assert torch.equal(x.grad, gx_ref) or torch.allclose(x.grad, gx_ref, rtol=1e-5, atol=1e-6)


def check_epilogue_forms(eng, dev):
    """ggl_spmm_epi_ex / ggl_segment_epi / ggl_gather_rows_f32_ex: sum and mean with "+ add + bias -> ReLU"
    applied in the aggregate's store (SAGEConv: mean + fc_self(x_dst) + bias -> act, sage_conv.py:100-108) ==
    the aggregate followed by the adds in torch, values and every gradient bit for bit; column blocks of a wider
    matrix assemble the full-width result (same dropout mask); the send-row gather == index_select."""
    g = torch.Generator(device="cpu").manual_seed(41)
    old = eng.chunk
    old_deg = int(eng.lib.ggl_get_option(b"col_block_min_degree"))
    eng.set_option("col_block_min_degree", 0)     # the wide cases run as column blocks whatever the graph's degree
    try:
        for chunk in (0, 8):
            eng.chunk = chunk
            eng.graph_cache.clear(); eng.seg_cache.clear()
            for (Nd, Ns, E, K) in ((40, 70, 600, 8), (64, 64, 900, 64), (50, 90, 700, 256), (6, 9, 0, 12), (45, 60, 500, 70)):
                ei = torch.stack([torch.randint(0, Ns, (E,), generator=g), torch.randint(0, Nd, (E,), generator=g)])
                if E:
                    ei[1, : E // 3] = 3
                ei = ei.to(dev)
                gp = eng.graph_plan(ei, Nd, Ns)
                go = torch.randn(Nd, K, generator=g).to(dev)
                for reduce in ("sum", "mean"):
                    for relu in (False, True):
                        mk = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_(True)  # noqa: E731
                        x, add, bias = mk(Ns, K), mk(Nd, K), mk(1, K)
                        x2, add2, bias2 = (t.detach().clone().requires_grad_(True) for t in (x, add, bias))
                        y = eng.spmm_epi(gp, None, x, reduce, add=add, bias=bias, relu=relu)
                        ref = eng.spmm(gp, None, x2, reduce) + add2
                        ref = ref + bias2
                        ref = torch.relu(ref) if relu else ref
                        assert torch.equal(y, ref), (chunk, Nd, E, K, reduce, relu)
                        if Nd * K:
                            y.backward(go)
                            ref.backward(go)
                            assert torch.equal(x.grad, x2.grad) and torch.equal(add.grad, add2.grad)
                            torch.testing.assert_close(bias.grad, bias2.grad, rtol=1e-5, atol=1e-5)
                        # the segment route on pre-gathered messages
                        m1 = x.detach()[ei[0]].clone().requires_grad_(True)
                        m2 = m1.detach().clone().requires_grad_(True)
                        a1, a2 = add.detach().clone().requires_grad_(True), add.detach().clone().requires_grad_(True)
                        ys = eng.segment_epi(m1, ei[1].contiguous(), Nd, reduce, add=a1, bias=bias.detach(), relu=relu)
                        seg = eng.c_segment_mean if reduce == "mean" else eng.c_segment_sum
                        rs = seg(m2, ei[1].contiguous(), Nd) + a2
                        rs = rs + bias.detach()
                        rs = torch.relu(rs) if relu else rs
                        assert torch.equal(ys, rs), (chunk, Nd, E, K, reduce, relu, "segment")
                        if Nd * K and E:
                            ys.backward(go)
                            rs.backward(go)
                            assert torch.equal(m1.grad, m2.grad) and torch.equal(a1.grad, a2.grad)
        eng.chunk = old
        eng.graph_cache.clear(); eng.seg_cache.clear()
        # column blocks: two edge sets added block by block, epilogue (with dropout) on the last one == full width
        N, E, K = 60, 800, 32
        e1 = torch.randint(0, N, (2, E), generator=g).to(dev)
        e2 = torch.randint(0, N, (2, E // 2), generator=g).to(dev)
        g1, g2 = eng.graph_plan(e1, N), eng.graph_plan(e2, N)
        w1, w2 = torch.rand(E, generator=g).to(dev), torch.rand(E // 2, generator=g).to(dev)
        x = torch.randn(N, K, generator=g).to(dev)
        bias = torch.randn(K, generator=g).to(dev)
        for p in (0.0, 0.4):
            rng = eng._rng_state(dev)
            st = rng.clone()
            full = torch.empty(N, K, device=dev)
            eng.spmm_sum_into(g1.fwd, g1.col, w1, x, full)
            eng.spmm_epi_into(g2.fwd, g2.col, w2, x, full, accumulate=True, bias=bias, relu=True, p_drop=p, rng=rng,
                              epi_K=K)
            rng.copy_(st)
            blk = torch.empty(N, K, device=dev)
            eng.spmm_sum_into(g1.fwd, g1.col, w1, x, blk)
            for c0 in (0, 8, 16, 24):
                eng.spmm_epi_into(g2.fwd, g2.col, w2, x[:, c0:c0 + 8], blk[:, c0:c0 + 8], accumulate=True, bias=bias,
                                  relu=True, p_drop=p, rng=rng, epi_K=K, col0=c0, advance_rng=(c0 == 24))
            assert torch.equal(full, blk), p
            assert torch.equal(rng, st + torch.tensor([0, 1 if p > 0 else 0], device=dev))
            rng.copy_(st)
            two = eng.bias_act(eng.spmm(g1, w1, x) + 0, None)  # (plain composition for the values)
            ref = torch.empty(N, K, device=dev)
            eng.spmm_sum_into(g1.fwd, g1.col, w1, x, ref)
            eng.spmm_sum_into(g2.fwd, g2.col, w2, x, ref, accumulate=True)
            rng.copy_(st)
            ref = eng.bias_act(ref, bias.reshape(1, K), relu=True, p_drop=p)
            assert torch.equal(full, ref) and two.shape == ref.shape
        # gather of send rows out of a column block
        src = torch.randn(50, 40, generator=g).to(dev)
        idx = torch.randint(0, 50, (77,), generator=g).to(dev)
        for (c0, c1) in ((0, 40), (8, 24), (3, 10)):
            out = torch.empty(77, c1 - c0, device=dev)
            eng.gather_rows_into(src[:, c0:c1], idx, out)
            assert torch.equal(out, src[:, c0:c1].index_select(0, idx))
        wide = torch.zeros(77, 64, device=dev)
        eng.gather_rows_into(src[:, 8:24], idx, wide[:, 32:48])
        assert torch.equal(wide[:, 32:48], src[:, 8:24][idx]) and float(wide[:, :32].abs().sum()) == 0
    finally:
        eng.chunk = old
        eng.set_option("col_block_min_degree", old_deg)
        eng.graph_cache.clear(); eng.seg_cache.clear()


def check_weight_dtype_guard(eng, dev):
    """A float64 / mis-shaped edge weight must never reach a kernel as a raw f32 pointer (it produced values
    off by 1e38): the explicit-plan ops raise like the reference's data_ptr<float>() would, and GCNConv with a
    float64 edge_weight takes the message() route, which promotes exactly as the reference layer does."""
    import pytest

    g = torch.Generator(device="cpu").manual_seed(51)
    N, E, K = 30, 200, 8
    ei = torch.randint(0, N, (2, E), generator=g).to(dev)
    x = torch.randn(N, K, generator=g).to(dev)
    w64 = torch.rand(E, generator=g, dtype=torch.float64).to(dev)
    gp = eng.graph_plan(ei, N)
    for call in (lambda: eng.spmm(gp, w64, x), lambda: eng.spmm_bias_act(gp, w64, x, None),
                 lambda: eng.spmm(gp, torch.rand(E, K).to(dev), x), lambda: eng.spmm_bias_act(gp, torch.rand(E + 1).to(dev), x)):
        with pytest.raises(RuntimeError):
            call()


def check_block_sampler(eng, dev, oracle):
    """The static-shape sampler (ggl_sample_hop / ggl_block_transpose / BlockMeanEpi): with a fan-out >= the
    largest degree it IS sample_adj without sampling -> bit-for-bit vs the restated reference (duplicate seeds
    included); sampled hops through the reference's invariants; padding rows empty; scratch restored; the block
    aggregate and its backward vs the written-out formula; draws of consecutive calls independent."""
    from gammagl_amd import sampler
    from gammagl_amd.layers import GraphSAGESampleModel

    rng = np.random.default_rng(5)
    N, E = 300, 2400
    ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
    ei[1, :64] = 7                                     # one heavy row (deg >= 64)
    ei = np.unique(ei, axis=1)                         # no multi-edges: distinct positions = distinct neighbours
    order = np.argsort(ei[1], kind="stable")
    rowptr = np.concatenate(([0], np.cumsum(np.bincount(ei[1], minlength=N)))).astype(np.int64)
    col = ei[0][order]
    deg = np.diff(rowptr)
    eit = to_t(ei, dev)
    seeds = np.array([7, 3, 9, 3, 100, 42, 7, 250], np.int64)     # duplicates stay, as in sample.cpp:24-29
    # (1) fan-out >= max degree: the deterministic branch, bit for bit
    big = int(deg.max())
    bs = sampler.BlockSampler(eit, [big], num_nodes=N, eng=eng)
    n_id, (blk,), counts = bs.sample(to_t(seeds, dev))
    ref_rp, ref_col, ref_nid, ref_eid = oracle.sample_adj_full(rowptr, col, seeds)
    nn, ne = (int(v) for v in to_np(counts)[:2])
    assert nn == len(ref_nid) and ne == len(ref_col)
    assert_same(to_np(blk.rowptr), ref_rp, "block rowptr")
    assert_same(to_np(blk.col)[:ne].astype(np.int64), ref_col, "block col")
    assert_same(to_np(n_id)[:nn], ref_nid, "block n_id")
    assert_same(to_np(blk.e_pos)[:ne], ref_eid, "block e_pos")
    assert (to_np(n_id)[nn:] == 0).all() and (to_np(blk.col)[ne:] == 0).all()
    assert bool((bs._first_pos == sampler._BIG).all())            # scratch handed back clean
    # fewer valid seeds than capacity: the rest of the rows are empty
    n_id2, (blk2,), c2 = bs.sample(to_t(seeds, dev), n_seeds=torch.tensor([3], device=dev))
    r2 = oracle.sample_adj_full(rowptr, col, seeds[:3])
    assert int(c2[0]) == len(r2[2]) and int(c2[1]) == len(r2[1])
    assert_same(to_np(blk2.rowptr)[:4], r2[0], "short rowptr") and None
    assert (to_np(blk2.rowptr)[3:] == r2[0][-1]).all()
    assert_same(to_np(n_id2)[: len(r2[2])], r2[2], "short n_id")
    # (2) sampled hops: invariants of sample.cpp (min(deg, fanout) distinct neighbours, rows sorted, seeds first)
    bs = sampler.BlockSampler(eit, [5, 3], num_nodes=N, eng=eng)
    sd = np.concatenate(([7], rng.permutation(N)[:15])).astype(np.int64)
    sd = sd[np.sort(np.unique(sd, return_index=True)[1])]
    n_id, blocks, counts = bs.sample(to_t(sd, dev))
    assert [b.fanout for b in blocks] == [3, 5] and blocks[1].n_dst_cap == len(sd)
    assert blocks[0].n_dst_cap == blocks[1].n_src_cap and n_id.shape[0] == blocks[0].n_src_cap
    for blk in blocks[::-1]:                                      # innermost first, as sampled
        nn, ne = (int(v) for v in to_np(blk.counts)[:2])
        nv = int(to_np(blk.n_seeds)[0])
        rp, cl, ep = to_np(blk.rowptr), to_np(blk.col).astype(np.int64), to_np(blk.e_pos)
        hop_seeds, hop_nid = to_np(blk.seeds), to_np(blk.n_id)
        k = np.diff(rp)
        assert (k[:nv] == np.minimum(deg[hop_seeds[:nv]], blk.fanout)).all() and (k[nv:] == 0).all() and rp[-1] == ne
        assert (hop_nid[:nv] == hop_seeds[:nv]).all() and len(np.unique(hop_nid[:nn])) == nn and (hop_nid[nn:] == 0).all()
        assert (cl[:ne] < nn).all() and (hop_nid[cl[:ne]] == col[ep[:ne]]).all()
        for i in range(nv):
            s0 = int(hop_seeds[i])
            pos = ep[rp[i]:rp[i + 1]]
            assert ((pos >= rowptr[s0]) & (pos < rowptr[s0 + 1])).all() and len(np.unique(pos)) == len(pos)
            assert (np.diff(cl[rp[i]:rp[i + 1]]) > 0).all()
    assert blocks[0].seeds is blocks[1].n_id and torch.equal(blocks[0].n_id, n_id)
    assert int(to_np(blocks[0].n_seeds)[0]) == int(to_np(blocks[1].counts)[0])
    assert bool((bs._first_pos == sampler._BIG).all())
    # (3) the block aggregate + fused epilogue and its backward (device-built CSC) vs the formula
    bs = sampler.BlockSampler(eit, [6], num_nodes=N, eng=eng)
    n_id, (blk,), counts = bs.sample(to_t(sd, dev))
    nn, ne = (int(v) for v in to_np(counts)[:2])
    K = 8
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(blk.n_src_cap, K, generator=g).to(dev).requires_grad_(True)
    add = torch.randn(blk.n_dst_cap, K, generator=g).to(dev).requires_grad_(True)
    bias = torch.randn(1, K, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(blk.n_dst_cap, K, generator=g).to(dev)
    y = eng.block_mean_epi(xs, blk, add=add, bias=bias, relu=True)
    y.backward(go)
    rp, cl = blk.rowptr, blk.col[:ne].long()
    rows = torch.repeat_interleave(torch.arange(blk.n_dst_cap, device=dev), rp[1:] - rp[:-1])
    xr, ar, br = (t.detach().clone().requires_grad_(True) for t in (xs, add, bias))
    cnt = (rp[1:] - rp[:-1]).clamp(min=1).unsqueeze(1).float()
    ref = torch.relu(torch.zeros(blk.n_dst_cap, K, device=dev).index_add_(0, rows, xr[cl]) / cnt + ar + br)
    ref.backward(go)
    torch.testing.assert_close(y.detach(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xs.grad, xr.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(add.grad, ar.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bias.grad, br.grad, rtol=1e-5, atol=1e-5)
    planT, dstT = blk.transposed()
    rpT = to_np(planT.rowptr)
    cln = to_np(cl)
    assert rpT[-1] == ne and (np.diff(rpT) == np.bincount(cln, minlength=blk.n_src_cap)).all()
    o = np.argsort(cln, kind="stable")
    assert (to_np(dstT)[:ne] == to_np(rows)[o]).all()
    # (4) consecutive calls draw independently (fan-out 4 of 64+ neighbours: ~4 * 4 / deg shared picks expected)
    bs = sampler.BlockSampler(eit, [4], num_nodes=N, eng=eng)
    s7 = torch.full((1500,), 7, dtype=torch.int64, device=dev)
    _, (b1,), _ = bs.sample(s7)
    _, (b2,), _ = bs.sample(s7)
    p1, p2 = to_np(b1.e_pos).reshape(1500, 4), to_np(b2.e_pos).reshape(1500, 4)
    shared = np.mean([len(set(a) & set(b)) for a, b in zip(p1, p2)])
    identical = sum(set(a) == set(b) for a, b in zip(p1, p2))
    assert shared < 0.5 and identical <= 3, (shared, identical)       # 0.25 expected; the XOR layout gave 1.55 / 57
    cnt7 = np.bincount(p1.reshape(-1) - rowptr[7], minlength=deg[7])
    assert np.abs(cnt7 - 6000 / deg[7]).max() < 6 * np.sqrt(6000 / deg[7])
    # (4b) capacities below the worst case: calibrated ones fit (no overflow, same invariants); too small ones
    # raise the overflow flag and stay memory-safe (rows cut at the edge capacity, ids below the node capacity)
    bs = sampler.BlockSampler(eit, [5, 3], num_nodes=N, eng=eng)
    caps = bs.calibrate(16, trials=6, slack=1.25)
    worst = bs.capacities(16)
    assert all(c[0] <= w[0] and c[1] <= w[1] for c, w in zip(caps, worst)) and caps[1][0] < worst[1][0]
    before = bs.overflow_count()
    n_id, blocks, _ = bs.sample(to_t(sd, dev), caps=caps)
    assert bs.overflow_count() == before and n_id.shape[0] == caps[1][0]
    assert blocks[1].n_src_cap == caps[0][0] and blocks[1].e_cap == caps[0][1] and blocks[0].n_dst_cap == caps[0][0]
    for blk in blocks:
        nn, ne, ov = (int(v) for v in to_np(blk.counts))
        assert ov == 0 and nn <= blk.n_src_cap and ne <= blk.e_cap and int(to_np(blk.rowptr)[-1]) == ne
        assert (to_np(blk.n_id)[to_np(blk.col).astype(np.int64)[:ne]] == col[to_np(blk.e_pos)[:ne]]).all()
    tiny = [(len(sd) + 8, 24), (len(sd) + 8 + 16, 40)]
    n_id, blocks, _ = bs.sample(to_t(sd, dev), caps=tiny)
    assert bs.overflow_count() > before
    for blk in blocks:
        nn, ne, ov = (int(v) for v in to_np(blk.counts))
        rp = to_np(blk.rowptr)
        assert nn <= blk.n_src_cap and ne <= blk.e_cap and rp[-1] == ne and (np.diff(rp) >= 0).all()
        assert (to_np(blk.col) >= 0).all() and (to_np(blk.col) < blk.n_src_cap).all()
    assert any(int(to_np(b.counts)[2]) == 1 for b in blocks) and bool((bs._first_pos == sampler._BIG).all())
    # (4c) a seed outside [0, N) is memory-safe: it gets no neighbours (no out-of-bounds read of rowptr or of the
    # relabel scratch) and stays in n_id, where the caller's feature gather reports it
    bad = to_t(np.array([7, N + 5, -1, 9], np.int64), dev)
    n_idb, blocks_b, cb = bs.sample(bad)
    bb = blocks_b[-1]                                             # the seeds' own block
    rpb = to_np(bb.rowptr)
    assert rpb[2] == rpb[1] and rpb[3] == rpb[2] and (to_np(bb.n_id)[:4] == to_np(bad)).all()
    assert bool((bs._first_pos == sampler._BIG).all())
    gotb = sampler.sample_adj(to_t(rowptr, dev), to_t(col, dev), bad, 3, eng=eng)
    assert to_np(gotb[0])[2] == to_np(gotb[0])[1] and (to_np(gotb[2])[:4] == to_np(bad)).all()
    # (5) the model runs on blocks end to end (shapes by capacity, loss on the seed rows)
    bs = sampler.BlockSampler(eit, [5, 3], num_nodes=N, eng=eng)
    n_id, blocks, _ = bs.sample(to_t(sd, dev))
    net = GraphSAGESampleModel(10, 8, 4, 0.0, 2).to(dev)
    xall = torch.randn(N, 10, generator=g).to(dev)
    out = net(xall.index_select(0, n_id), blocks)
    assert out.shape == (blocks[1].n_dst_cap, 4) and bool(torch.isfinite(out).all())
    out[: len(sd)].sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())


# --------------------------------------------------------------------------------------------------
# the dgNN boundary (row G's call site)
# the one statement through which GammaGL's fused GAT layer binds its kernel (layers/conv/fusedgat_conv.py:70-71)
DGNN_BINDING = "from dgNN.operators import GATConvFuse\nop = GATConvFuse\n"


def dgnn_standin(tmp_path):
    """A stand-in module holding exactly the layer's import, with gammagl_amd/compat/dgNN reachable as `dgNN` (a
    symlink in a scratch directory on sys.path: what INTEGRATION.md tells a maintainer to do).  No reference file is
    copied."""
    import importlib
    import os
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = tmp_path / "dgnn_site"
    root.mkdir(exist_ok=True)
    link = root / "dgNN"
    if not link.exists():
        os.symlink(os.path.join(repo, "gammagl_amd", "compat", "dgNN"), link)
    (root / "fused_layer_standin.py").write_text(DGNN_BINDING)
    sys.path.insert(0, str(root))
    try:
        for m in ("dgNN", "dgNN.operators", "fused_layer_standin"):
            sys.modules.pop(m, None)
        return importlib.import_module("fused_layer_standin")
    finally:
        sys.path.pop(0)


def check_dgnn_dropin(dev, oracle, tmp_path):
    """`GATConvFuse(alpha_dst, alpha_src, row_ptr, col_ind, col_ptr, row_ind, permute, slope, x, 0)` called exactly as
    fusedgat_conv.py:102-121 does — CSR built on edge_index[0] with the layer's own preprocessing steps restated in
    torch — equals the in-tree GATConv math (gat_conv.py:103-112 + softmax.py:29-35, the oracle) on the graph whose
    destinations are edge_index[0]: forward 1e-5, the three gradients 1e-4 (the GAT tolerances of this suite)."""
    mod = dgnn_standin(tmp_path)
    rng = np.random.default_rng(23)
    for N, E, H, C in ((40, 300, 4, 8), (64, 900, 8, 8), (30, 200, 2, 5), (50, 400, 8, 41)):
        ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
        ei[0, : E // 5] = 3                                   # a hub row
        x = rng.standard_normal((N, H, C)).astype(np.float32)
        a_dst = rng.standard_normal((N, H)).astype(np.float32)
        a_src = rng.standard_normal((N, H)).astype(np.float32)
        go = rng.standard_normal((N, H, C)).astype(np.float32)
        # fusedgat_conv.py:106-117: sort by row, CSR pointer, then sort by column carrying the position -> CSC + permute
        eit = torch.from_numpy(ei)
        o = torch.argsort(eit[0] * N + eit[1], stable=True)
        er_sorted = eit[:, o]
        row_ptr = torch.zeros(N + 1, dtype=torch.int64)
        row_ptr[1:] = torch.cumsum(torch.bincount(er_sorted[0], minlength=N), 0)
        col_ind = er_sorted[1]
        permute = torch.arange(E)
        o2 = torch.argsort(er_sorted[1] * N + er_sorted[0], stable=True)
        ec_sorted, permute = er_sorted[:, o2], permute[o2]
        row_ind = ec_sorted[0]
        col_ptr = torch.zeros(N + 1, dtype=torch.int64)
        col_ptr[1:] = torch.cumsum(torch.bincount(ec_sorted[1], minlength=N), 0)
        i32 = lambda t: t.to(torch.int32).to(dev)             # noqa: E731  (tlx.convert_to_tensor(.., tlx.int32), :113-117)
        xt, ad, as_ = (to_t(v, dev).requires_grad_(True) for v in (x, a_dst, a_src))
        out = mod.op(ad, as_, i32(row_ptr), i32(col_ind), i32(col_ptr), i32(row_ind), i32(permute), 0.2, xt, 0.0)
        out.backward(to_t(go, dev))
        # oracle: edges j -> i with i = edge_index[0] (the aggregating row), j = edge_index[1]; el = the source's term
        index = np.stack([ei[1], ei[0]])
        want = oracle.gat_fwd(index, a_src, a_dst, x, 0.2)
        gel, ger, gx = oracle.gat_bwd(index, a_src, a_dst, x, go, 0.2)
        np.testing.assert_allclose(to_np(out), want, rtol=1e-5, atol=1e-5, err_msg=f"GATConvFuse forward {H}x{C}")
        np.testing.assert_allclose(to_np(xt.grad), gx, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(to_np(as_.grad), gel, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(to_np(ad.grad), ger, rtol=1e-4, atol=1e-4)
        # a second call with the same five tensors hits the plan cache (identity + version of row_ptr)
    # dropout: kept coefficients are rescaled, dropped ones are zero — the op applies attn_drop as passed
    N, E, H, C = 32, 256, 2, 4
    ei = torch.from_numpy(rng.integers(0, N, size=(2, E)).astype(np.int64))
    o = torch.argsort(ei[0] * N + ei[1], stable=True)
    e1 = ei[:, o]
    rp = torch.zeros(N + 1, dtype=torch.int64); rp[1:] = torch.cumsum(torch.bincount(e1[0], minlength=N), 0)
    o2 = torch.argsort(e1[1] * N + e1[0], stable=True)
    e2 = e1[:, o2]
    cp = torch.zeros(N + 1, dtype=torch.int64); cp[1:] = torch.cumsum(torch.bincount(e2[1], minlength=N), 0)
    args = [t.to(torch.int32).to(dev) for t in (rp, e1[1], cp, e2[0], torch.arange(E)[o2])]
    z = torch.zeros(N, H, device=dev)
    ones = torch.ones(N, H, C, device=dev)
    y0 = mod.op(z, z, *args, 0.2, ones, 0.0)
    has = (rp[1:] > rp[:-1]).to(dev)
    assert float((y0[has] - 1).abs().max()) < 1e-5            # softmax weights sum to one
    y5 = mod.op(z, z, *args, 0.2, ones, 0.5)
    assert not torch.allclose(y5, y0) and bool(torch.isfinite(y5).all())


def check_neighbor_sample(fn, dev, oracle):
    """`cuda_torch_neighbor_sample(colptr, row, input_nodes, fanouts, replace, directed, seed)` (neighbor_sample.cu:744-778):
    with fan-out -1 every array equals the Python restatement of cu_neighbor_sample (oracle.neighbor_sample_full: node
    order, edge order, local positions); with positive fan-outs the contract of every hop: min(degree, fan-out) DISTINCT
    in-edges of each frontier node, new nodes appended in ascending order, local positions consistent."""
    rng = np.random.default_rng(31)
    for N, E, seeds, hops in ((30, 200, [3, 7, 11], 2), (200, 3000, list(range(0, 40, 3)), 3), (50, 60, [49, 0], 4), (10, 0, [1, 2], 2)):
        src = rng.integers(0, N, size=E)
        dst = np.sort(rng.integers(0, N, size=E))
        colptr = np.zeros(N + 1, np.int64)
        np.add.at(colptr, dst + 1, 1)
        colptr = np.cumsum(colptr)
        row = src.astype(np.int64)
        cp, rw, sd = to_t(colptr, dev), to_t(row, dev), to_t(np.array(seeds, np.int64), dev)
        got = fn(cp, rw, sd, torch.tensor([-1] * hops), False, False, 0)
        want = oracle.neighbor_sample_full(colptr, row, seeds, hops)
        assert isinstance(got, list) and len(got) == 4
        for g, w, nm in zip(got, want, ("cols", "rows", "nodes", "edges")):
            assert g.dtype == torch.int64
            np.testing.assert_array_equal(to_np(g), w, err_msg=f"neighbor_sample -1: {nm} (N={N})")
        # positive fan-outs: the hop contract
        fan = [3, 2, 2, 2][:hops]
        cols, rows, nodes, edges = (to_np(t) for t in fn(cp, rw, sd, torch.tensor(fan), False, False, 0))
        assert nodes[: len(seeds)].tolist() == seeds and len(set(nodes.tolist())) == len(nodes)
        assert len(cols) == len(rows) == len(edges)
        if len(edges):
            assert np.array_equal(nodes[rows], row[edges]), "far end of every edge sits at its local position"
            owner = nodes[cols]
            assert np.all((colptr[owner] <= edges) & (edges < colptr[owner + 1])), "every edge belongs to its owner's stretch"
        # per owner: min(deg, fanout of the hop it was sampled in) distinct edges; owners appear hop by hop
        at, f_lo, f_hi = 0, 0, len(seeds)
        for f in fan:
            exp = np.minimum(colptr[nodes[f_lo:f_hi] + 1] - colptr[nodes[f_lo:f_hi]], f)
            n_e = int(exp.sum())
            hop_cols, hop_edges = cols[at:at + n_e], edges[at:at + n_e]
            assert np.array_equal(hop_cols, np.repeat(np.arange(f_lo, f_hi), exp)), "edges of a hop: frontier node by node"
            for k in range(f_lo, f_hi):
                mine = hop_edges[hop_cols == k]
                assert len(set(mine.tolist())) == len(mine), "distinct in-edges"
            at += n_e
            new = np.setdiff1d(np.unique(row[hop_edges]), nodes[:f_hi])
            if len(new) == 0:
                break
            assert np.array_equal(nodes[f_hi:f_hi + len(new)], new), "new nodes ascending"
            f_lo, f_hi = f_hi, f_hi + len(new)
        assert at == len(edges)
    # DUPLICATE input nodes (the reference never deduplicates them: get_new_input_nodes only compares NEW ids with the list,
    # neighbor_sample.cu:436-532): every copy is a frontier entry of its own, sampled independently, and an edge that leads
    # to a duplicated node points at ONE of its positions — which one is unpinned (kernal_get_row's search is not
    # restated; here: the first).  What is pinned: positions are valid and hold the right node, owners are per entry.
    N, E = 40, 400
    src = rng.integers(0, N, size=E)
    dst = np.sort(rng.integers(0, N, size=E))
    colptr = np.zeros(N + 1, np.int64)
    np.add.at(colptr, dst + 1, 1)
    colptr = np.cumsum(colptr)
    row = src.astype(np.int64)
    seeds = [5, 9, 5, 5, 12]
    cp, rw, sd = to_t(colptr, dev), to_t(row, dev), to_t(np.array(seeds, np.int64), dev)
    for fan in ([-1, -1], [3, 2]):
        cols, rows, nodes, edges = (to_np(t) for t in fn(cp, rw, sd, torch.tensor(fan), False, False, 0))
        assert nodes[: len(seeds)].tolist() == seeds
        assert len(set(nodes[len(seeds):].tolist())) == len(nodes) - len(seeds), "appended nodes are distinct"
        assert not (set(nodes[len(seeds):].tolist()) & set(seeds)), "a seed is never appended again"
        assert np.array_equal(nodes[rows], row[edges]) and rows.min() >= 0 and rows.max() < len(nodes)
        owner = nodes[cols]
        assert np.all((colptr[owner] <= edges) & (edges < colptr[owner + 1]))
        f = fan[0]
        deg = colptr[np.array(seeds) + 1] - colptr[np.array(seeds)]
        exp = deg if f < 0 else np.minimum(deg, f)
        n0 = int(exp.sum())
        assert np.array_equal(cols[:n0], np.repeat(np.arange(len(seeds)), exp)), "every copy of a seed is sampled for, in place"
        assert np.all(rows[np.isin(row[edges], [5])] == 0), "an edge leading to a duplicated seed points at its first position"


def check_cpp_fused_route(ops, eng, dev, oracle=None):
    """torch.ops.ggl.{gat_fused, gat_fused_csr, bias_act, spmm_epi, segment_epi, sample_hop} (registered from C++,
    csrc/torch/ggl_torch.cpp) against the ctypes Engine on the same kernel library: values and every gradient equal
    bit for bit (same kernels, same launch policy), dropout included when both RNG states are seeded alike."""
    g = torch.Generator().manual_seed(41)
    N, E = 90, 2500
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :700] = 5                                   # a hub row (longer than the plan's threshold of 256)
    ei = ei.to(dev)

    def tri(H, C):
        return (torch.randn(N, H, C, generator=g).to(dev), torch.randn(N, H, generator=g).to(dev),
                torch.randn(N, H, generator=g).to(dev), torch.randn(N, H, C, generator=g).to(dev))

    for H, C, p in ((4, 8, 0.0), (8, 8, 0.0), (2, 5, 0.0), (8, 41, 0.0), (3, 12, 0.0), (4, 8, 0.4), (2, 5, 0.3)):
        x, el, er, go = tri(H, C)
        res = []
        for which in ("engine", "cpp"):
            torch.manual_seed(1234)
            eng.reseed() if which == "engine" else ops.reseed()
            xa, ela, era = (t.clone().requires_grad_(True) for t in (x, el, er))
            if which == "engine":
                y = eng.gat_fused(ei, ela, era, xa, 0.2, N, p, True)
            else:
                y = ops.gat_fused(ei, ela, era, xa, 0.2, N, p)
            y.backward(go)
            res.append((y.detach(), xa.grad, ela.grad, era.grad))
        for a, b, nm in zip(res[0], res[1], ("out", "gx", "gel", "ger")):
            assert torch.equal(a, b), f"gat_fused {H}x{C} p={p}: {nm} differs from the engine's"
    # the caller's CSR (dgNN's argument list): rows aggregate, no sort
    gp = eng.graph_plan(ei, N)
    five = (gp.fwd.rowptr.clone(), gp.col.clone(), gp.bwd.rowptr.clone(), gp.colT.clone(), gp.posT.clone())
    x, el, er, go = tri(4, 8)
    xa, ela, era = (t.clone().requires_grad_(True) for t in (x, el, er))
    ya = eng.gat_fused(eng.graph_plan_from_csr(*five), ela, era, xa, 0.2, N, 0.0, True)
    ya.backward(go)
    for call in range(2):                              # the second call hits the C++ side's CSR plan cache
        xb, elb, erb = (t.clone().requires_grad_(True) for t in (x, el, er))
        yb = ops.gat_fused_csr(*five, elb, erb, xb, 0.2, 0.0)
        yb.backward(go)
        assert torch.equal(ya.detach(), yb.detach()) and torch.equal(xa.grad, xb.grad) and torch.equal(ela.grad, elb.grad) \
            and torch.equal(era.grad, erb.grad), f"gat_fused_csr call {call}"
    # the layer epilogue alone
    a = torch.randn(N, 24, generator=g).to(dev)
    b = torch.randn(24, generator=g).to(dev)
    go2 = torch.randn(N, 24, generator=g).to(dev)
    for bias, relu, p in ((b, True, 0.0), (None, True, 0.0), (b, False, 0.0), (b, True, 0.5), (None, False, 0.25)):
        res = []
        for which in ("engine", "cpp"):
            torch.manual_seed(77)
            eng.reseed() if which == "engine" else ops.reseed()
            aa = a.clone().requires_grad_(True)
            bb = bias.clone().requires_grad_(True) if bias is not None else None
            y = eng.bias_act(aa, bb, relu, p, True) if which == "engine" else ops.bias_act(aa, bb, relu, p)
            y.backward(go2)
            res.append((y.detach(), aa.grad, bb.grad if bb is not None else None))
        for u, v in zip(res[0], res[1]):
            assert (u is None and v is None) or torch.equal(u, v), ("bias_act", relu, p)
    # ... and in the aggregate's store
    w = torch.rand(E, generator=g).to(dev)
    for K, mean, use_add, use_b, relu, p in ((24, False, False, True, True, 0.0), (24, True, True, True, True, 0.0),
                                               (24, False, True, False, False, 0.3), (7, False, True, True, True, 0.0),
                                               (7, True, False, True, False, 0.0)):
        xk = torch.randn(N, K, generator=g).to(dev)
        addk = torch.randn(N, K, generator=g).to(dev) if use_add else None
        bk = torch.randn(K, generator=g).to(dev) if use_b else None
        gok = torch.randn(N, K, generator=g).to(dev)
        res = []
        for which in ("engine", "cpp"):
            torch.manual_seed(99)
            eng.reseed() if which == "engine" else ops.reseed()
            xx = xk.clone().requires_grad_(True)
            ad = addk.clone().requires_grad_(True) if addk is not None else None
            bb = bk.clone().requires_grad_(True) if bk is not None else None
            if which == "engine":
                y = eng.spmm_epi(eng.graph_plan(ei, N), w, xx, "mean" if mean else "sum", ad, bb, relu, p, True)
            else:
                y = ops.spmm_epi(ei, w, xx, mean, ad, bb, relu, p)
            y.backward(gok)
            res.append((y.detach(), xx.grad, ad.grad if ad is not None else None, bb.grad if bb is not None else None))
        for u, v in zip(res[0], res[1]):
            assert (u is None and v is None) or torch.equal(u, v), ("spmm_epi", K, mean, use_add, use_b, relu, p)
    msg = torch.randn(E, 12, generator=g).to(dev)
    ids = ei[1].contiguous()
    addn = torch.randn(N, 12, generator=g).to(dev)
    bn = torch.randn(12, generator=g).to(dev)
    gon = torch.randn(N, 12, generator=g).to(dev)
    for mean in (True, False):
        res = []
        for which in ("engine", "cpp"):
            m_ = msg.clone().requires_grad_(True)
            ad, bb = addn.clone().requires_grad_(True), bn.clone().requires_grad_(True)
            y = eng.segment_epi(m_, ids, N, "mean" if mean else "sum", ad, bb, True) if which == "engine" \
                else ops.segment_epi(m_, ids, N, mean, ad, bb, True)
            y.backward(gon)
            res.append((y.detach(), m_.grad, ad.grad, bb.grad))
        for u, v in zip(res[0], res[1]):
            assert torch.equal(u, v), ("segment_epi", mean)
    # one static-shape sampler hop: same kernels, same {seed, offset} -> the same block
    from gammagl_amd.ops import _ptr

    plan = eng.seg_plan(ids, N)
    rowptr = plan.rowptr
    col = ei[0] if plan.perm is None else ei[0][plan.perm.long()]
    seeds = torch.arange(0, 40, 3, device=dev)
    b_cap, fan = int(seeds.shape[0]), 4
    e_cap, s_cap = b_cap * fan, b_cap + b_cap * fan
    n_seeds = torch.full((1,), b_cap - 2, dtype=torch.int64, device=dev)     # two padding slots
    outs = []
    for which in ("engine", "cpp"):
        torch.manual_seed(5)
        first_pos = torch.full((N,), 1 << 62, dtype=torch.int64, device=dev)
        if which == "engine":
            eng.reseed()
            o_rp = torch.empty(b_cap + 1, dtype=torch.int64, device=dev)
            o_col = torch.empty(e_cap, dtype=torch.int32, device=dev)
            o_eid = torch.empty(e_cap, dtype=torch.int64, device=dev)
            o_nid = torch.empty(s_cap, dtype=torch.int64, device=dev)
            cnt = torch.empty(3, dtype=torch.int64, device=dev)
            wsb = eng.lib.ggl_sample_hop_workspace_bytes(b_cap, e_cap)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            eng._check(eng.lib.ggl_sample_hop(_ptr(rowptr), _ptr(col.contiguous()), _ptr(seeds), _ptr(n_seeds), b_cap, N, fan, e_cap,
                                              s_cap, _ptr(eng._rng_state(torch.device(dev))), _ptr(first_pos), _ptr(o_rp), _ptr(o_col),
                                              _ptr(o_eid), _ptr(o_nid), _ptr(cnt), _ptr(ws), wsb, eng._stream(torch.device(dev))))
            outs.append((o_rp, o_col, o_eid, o_nid, cnt))
        else:
            ops.reseed()
            outs.append(ops.sample_hop(rowptr, col.contiguous(), seeds, n_seeds, N, fan, e_cap, s_cap, first_pos))
        assert bool((first_pos == (1 << 62)).all()), "the relabel scratch comes back clean"
    nn, ne, ovf = outs[0][4].tolist()
    assert outs[1][4].tolist() == [nn, ne, ovf] and ovf == 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1][:ne], outs[1][1][:ne]) \
        and torch.equal(outs[0][2][:ne], outs[1][2][:ne]) and torch.equal(outs[0][3][:nn], outs[1][3][:nn])


def check_round4_paths(eng, dev, oracle):
    """Paths added in round 4: 16-byte lanes for int32 / int64 messages (K >= 16 / 8, aligned) and the zero-padded copy
    the gspmm-max walk takes for wide rows that are not 16-byte pieces (K > 128, K % 4 != 0) — values, argmax (through the
    gradient) and the backward walk, bit for bit against the oracle."""
    rng = np.random.default_rng(77)
    N, E = 37, 900
    ids = rng.integers(0, N, size=E).astype(np.int64)
    ids[:300] = 4
    it = to_t(ids, dev)
    for dt, widths in ((np.int32, (16, 64, 100, 17)), (np.int64, (8, 32, 50, 7))):
        for K in widths:
            x = rng.integers(-1000, 1000, size=(E, K)).astype(dt)
            xt = to_t(x, dev)
            assert_same(to_np(eng.c_segment_sum(xt, it, N)), oracle.segment_sum(x, ids, N), f"{dt.__name__} sum K{K}")
            assert_same(to_np(eng.c_segment_mean(xt, it, N)), oracle.segment_mean(x, ids, N), f"{dt.__name__} mean K{K}")
            mx, arg = eng.segment_max_with_arg(xt, it, N)
            omx, oarg = oracle.segment_max(x, ids, N)
            assert_same(to_np(mx), omx, f"{dt.__name__} max K{K}")
            assert_same(to_np(arg), oarg, f"{dt.__name__} argmax K{K}")
    N, E = 30, 600
    for K in (130, 201, 602):
        index = np.stack([rng.integers(0, N, size=E), rng.integers(0, N, size=E)]).astype(np.int64)
        index[1, :200] = 2
        w = rng.standard_normal(E).astype(np.float32)
        x = (rng.integers(-4, 5, size=(N, K)) * 0.25).astype(np.float32)        # ties for the argmax
        go = rng.standard_normal((N, K)).astype(np.float32)
        xt = to_t(x, dev).requires_grad_(True)
        y = eng.c_spmm_max(to_t(index, dev), to_t(w, dev), xt)
        y.backward(to_t(go, dev))
        oy, oarg = oracle.spmm_max_fwd(index, w, x)
        assert_same(to_np(y), oy, f"spmm max K{K} (padded walk)")
        np.testing.assert_allclose(to_np(xt.grad), oracle.spmm_max_bwd(index, w, go, oarg), rtol=1e-5, atol=1e-5)


def check_max_backward_forms(eng, dev, oracle, chunk=64):
    """gspmm(max) backward three ways — int64 witnesses (ggl_spmm_max_bwd), int32 witnesses (ggl_spmm_max_bwd32) and the
    1-bit winner mask of round 5 (ggl_spmm_max_mask in destination order -> ggl_spmm_max_bwd_mask in source order): the same
    gradient, bit for bit, as the oracle (spmm_max_cpu.cpp:57-99) — ties, DUPLICATE edges (each copy is fed), empty rows, a
    hub source (chunked transposed rows) and a hub destination (the mask kernel's chunk blocks), sorted and shuffled edge
    lists, widths below / at / above one mask word, ragged widths, several 256-column passes."""
    old = eng.chunk
    eng.chunk = chunk
    eng.clear_caches()
    try:
        rng = np.random.default_rng(11)
        N, E = 60, 3000
        for K in (1, 4, 7, 32, 33, 47, 64, 100, 128, 130, 160, 256, 300, 512, 600):
            index = np.stack([rng.integers(0, N - 4, size=E), rng.integers(0, N - 4, size=E)]).astype(np.int64)
            index[0, :900] = 2                                          # hub source
            index[1, 1200:2100] = 7                                     # hub destination
            index[:, 1000:1100] = index[:, 1100:1200]                   # duplicate edges
            for shuffle in (True, False):
                if shuffle:
                    index = np.ascontiguousarray(index[:, rng.permutation(E)])
                else:
                    index = np.ascontiguousarray(index[:, np.argsort(index[1], kind="stable")])
                w = rng.standard_normal(E).astype(np.float32)
                xs = np.round(rng.standard_normal((N, K)) * 2).astype(np.float32)   # (rounded: plenty of ties)
                go = rng.standard_normal((N, K)).astype(np.float32)
                _, arg = oracle.spmm_max_fwd(index, w, xs)
                want = oracle.spmm_max_bwd(index, w, go, arg)
                it, wt = to_t(index, dev), to_t(w, dev)
                gp = eng.graph_plan(it, N)
                assert gp.fwd.n_long >= 1 and gp.bwd.n_long >= 1
                one_piece = to_np(gp.bwd.counts() <= gp.bwd.chunk)          # (chunked transposed rows: within rounding)
                for name, opts in (("int64 witnesses", {"maxbwd_mask": 0, "maxbwd_arg32": 0}),
                                   ("int32 witnesses", {"maxbwd_mask": 0, "maxbwd_arg32": 1}),
                                   ("winner mask, forward order", {"maxbwd_mask": 1, "maxbwd_arg32": 0}),
                                   ("winner mask, scattered", {"maxbwd_mask": 1, "maxbwd_arg32": 0, "maxbwd_mask_scatter": 1})):
                    with option(eng, "maxbwd_mask", opts["maxbwd_mask"]), option(eng, "maxbwd_arg32", opts["maxbwd_arg32"]), \
                            option(eng, "maxbwd_mask_kmax", 0), \
                            option(eng, "maxbwd_mask_scatter", opts.get("maxbwd_mask_scatter", 0)):
                        for call in range(2):       # (second call: weights streamed from their sorted copy)
                            xt = to_t(xs, dev).requires_grad_(True)
                            eng.c_spmm_max(it, wt, xt).backward(to_t(go, dev))
                            got = to_np(xt.grad)
                            assert_same(got[one_piece], want[one_piece], f"max backward, {name}, K{K} shuffle={shuffle} call {call}")
                            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)
                # no weights
                with option(eng, "maxbwd_mask", 1), option(eng, "maxbwd_mask_kmax", 0):    # (kmax 0: no upper bound on the width)
                    xt = to_t(xs, dev).requires_grad_(True)
                    eng.c_spmm_max(it, None, xt).backward(to_t(go, dev))
                    ones = np.ones(E, np.float32)
                    _, arg1 = oracle.spmm_max_fwd(index, ones, xs)
                    want1 = oracle.spmm_max_bwd(index, ones, go, arg1)
                    assert_same(to_np(xt.grad)[one_piece], want1[one_piece], f"max backward, winner mask, no weights K{K}")
    finally:
        eng.chunk = old
        eng.clear_caches()
