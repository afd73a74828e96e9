"""Pins the oracle (oracle/ggl_oracle.c) against the golden vectors the reference itself produced
(tests/golden/*.npz, made by tests/golden/make_golden.py) — bit-exact, CPU only."""
import numpy as np
import pytest

DT = ["uint8", "int8", "int16", "int32", "int64", "float16", "bfloat16", "float32", "float64"]
KAT_DT = ["int8", "int16", "int32", "int64", "float16", "float32", "float64"]


def same(a, b):
    """bitwise equality that treats NaN == NaN (payload-insensitive)."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        ok = (a == b) | (np.isnan(a) & np.isnan(b))
        # distinguish -0.0 / +0.0
        ok &= np.signbit(a) == np.signbit(b)
        return bool(ok.all())
    return bool((a == b).all())


@pytest.mark.parametrize("op", ["sum", "mean", "max"])
@pytest.mark.parametrize("dt", KAT_DT)
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_reference_kat_63(golden, oracle, op, dt, dim):
    g = golden["kat"]
    x, y = g[f"{op}_{dt}_d{dim}_x"], g[f"{op}_{dt}_d{dim}_y"]
    fn = {"sum": oracle.segment_sum, "mean": oracle.segment_mean,
          "max": lambda *a: oracle.segment_max(*a)[0]}[op]
    assert same(fn(x, g["idx"], 2), y)


def test_reference_message_passing_degree_softmax_norm(golden, oracle):
    g = golden["kat"]
    ei = g["mp_ei"]
    msg = g["mp_x"][ei[0]]
    assert same(oracle.segment_sum(msg, ei[1], 4), g["mp_sum"])
    assert same(oracle.segment_mean(msg, ei[1], 4), g["mp_mean"])
    assert same(oracle.segment_max(msg, ei[1], 4)[0], g["mp_max"])
    assert same(oracle.segment_sum(np.ones(5, np.int64), g["deg_row"], 3), g["deg_out"])
    # segment_softmax (utils/softmax.py:29-35)
    x_e = g["sm_x"]
    mx = oracle.segment_max(x_e, ei[1], 4)[0]
    ex = np.exp(x_e - mx[ei[1]], dtype=np.float32)
    den = oracle.segment_sum(ex, ei[1], 4)
    score = ex / (den[ei[1]] + np.float32(1e-16))
    np.testing.assert_allclose(score, g["sm_score"], rtol=1e-6)
    # calc_gcn_norm (utils/norm.py:24-30)
    e2 = g["norm_ei"]
    deg = oracle.segment_sum(np.ones((6, 1), np.float32), e2[0], 4).reshape(-1)
    dis = deg ** np.float32(-0.5)
    np.testing.assert_allclose(dis[e2[0]] * dis[e2[1]], g["norm_w"], rtol=1e-6)
    # docstrings (the C++ path leaves empty max rows at lowest(), not 0)
    assert same(oracle.segment_max(g["doc_x"], g["doc_ids"], 3)[0], g["doc_max"])
    assert same(oracle.spmm_sum_fwd(g["doc_gi"], 2 * np.ones(8, np.float32), 2 * np.ones((5, 8), np.float32)),
                g["doc_gspmm"])


@pytest.mark.parametrize("dt", DT)
def test_segment_all_dtypes_bit_exact(golden, oracle, dt):
    g = golden["segment"]
    bf = dt == "bfloat16"
    for ci in range(int(g["ncases"])):
        ids, N = g[f"c{ci}_ids"], int(g[f"c{ci}_N"])
        x = g[f"c{ci}_{dt}_x"]
        assert same(oracle.segment_sum(x, ids, N, bf16=bf), g[f"c{ci}_{dt}_sum"]), (ci, "sum")
        assert same(oracle.segment_mean(x, ids, N, bf16=bf), g[f"c{ci}_{dt}_mean"]), (ci, "mean")
        assert same(oracle.segment_max(x, ids, N, bf16=bf)[0], g[f"c{ci}_{dt}_max"]), (ci, "max")


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_segment_forward_backward(golden, oracle, dt):
    g = golden["segment"]
    for bi in range(int(g["nbwd"])):
        ids, N = g[f"b{bi}_ids"], int(g[f"b{bi}_N"])
        k = f"b{bi}_{dt}"
        x, go = g[k + "_x"], g[k + "_g"]
        assert same(oracle.segment_sum(x, ids, N), g[k + "_sum"])
        assert same(oracle.segment_mean(x, ids, N), g[k + "_mean"])
        mx, arg = oracle.segment_max(x, ids, N)
        assert same(mx, g[k + "_max"])
        assert same(oracle.segment_sum_bwd(go, ids, N), g[k + "_sum_gx"])
        assert same(oracle.segment_mean_bwd(go, ids, N), g[k + "_mean_gx"])
        # the gradient is the argmax witness: bit-exact means every argmax index agrees
        assert same(oracle.segment_max_bwd(go, arg, len(ids)), g[k + "_max_gx"])


def test_segment_max_nan_inf_signed_zero(golden, oracle):
    g = golden["segment"]
    mx, arg = oracle.segment_max(g["sp_x"], g["sp_ids"], 4)
    assert same(mx, g["sp_max"])
    go = np.arange(16, dtype=np.float32).reshape(4, 4) + 1
    assert same(oracle.segment_max_bwd(go, arg, 4), g["sp_gx"])
    assert same(oracle.segment_sum(g["sp_x"], g["sp_ids"], 4), g["sp_sum"])


@pytest.mark.parametrize("nm", ["f16", "bf16"])
def test_half_accumulation_and_count_saturation(golden, oracle, nm):
    g = golden["segment"]
    bf = nm == "bf16"
    x = g[f"sat_{nm}_x"]
    assert same(oracle.segment_sum(x, g["sat_ids"], 3, bf16=bf), g[f"sat_{nm}_sum"])
    assert same(oracle.segment_mean(x, g["sat_ids"], 3, bf16=bf), g[f"sat_{nm}_mean"])


def test_gspmm_forward_backward(golden, oracle):
    g = golden["spmm"]
    for ci in range(int(g["nspmm"])):
        k = f"s{ci}"
        idx, w, x, go = g[k + "_index"], g[k + "_w"], g[k + "_x"], g[k + "_g"]
        assert same(oracle.spmm_sum_fwd(idx, w, x), g[k + "_sum"])
        assert same(oracle.spmm_sum_bwd(idx, w, go), g[k + "_sum_gx"])
        ym, cnt = oracle.spmm_mean_fwd(idx, w, x)
        assert same(ym, g[k + "_mean"])
        assert same(oracle.spmm_mean_bwd(idx, w, go, cnt), g[k + "_mean_gx"])
        yx, arg = oracle.spmm_max_fwd(idx, w, x)
        assert same(yx, g[k + "_max"])
        assert same(oracle.spmm_max_bwd(idx, w, go, arg), g[k + "_max_gx"])


def test_bspmm_forward_backward(golden, oracle):
    g = golden["spmm"]
    for bi in range(int(g["nbspmm"])):
        k = f"bs{bi}"
        idx, w, x, go = g[k + "_index"], g[k + "_w"], g[k + "_x"], g[k + "_g"]
        assert same(oracle.bspmm_sum_fwd(idx, w, x), g[k + "_y"])
        gx, gw = oracle.bspmm_sum_bwd(idx, w, x, go)
        assert same(gx, g[k + "_gx"])
        assert same(gw, g[k + "_gw"])


def test_gcn_layer_fixture(golden, oracle):
    g = golden["layers"]
    ei, x, W, b = g["gcn_ei"], g["gcn_x"], g["gcn_W"], g["gcn_b"]
    N = x.shape[0]
    ones = np.ones(ei.shape[1], np.float32)
    wts = oracle.segment_sum(ones, ei[0], N) ** np.float32(-0.5)
    wts = wts[ei[0]] * (oracle.segment_sum(ones, ei[1], N) ** np.float32(-0.5))[ei[1]]
    np.testing.assert_allclose(wts, g["gcn_w"], rtol=1e-6)
    h = x @ W
    y = oracle.spmm_sum_fwd(ei, g["gcn_w"], h) + b
    np.testing.assert_allclose(y, g["gcn_y"], rtol=1e-5, atol=1e-6)
    gW = x.T @ oracle.spmm_sum_bwd(ei, g["gcn_w"], g["gcn_g"])
    np.testing.assert_allclose(gW, g["gcn_gW"], rtol=1e-4, atol=1e-5)


def test_gat_math_fixture(golden, oracle):
    g = golden["layers"]
    ei, x, el, er = g["gat_ei"], g["gat_x"], g["gat_el"], g["gat_er"]
    y, alpha = oracle.gat_fwd(ei, el, er, x, 0.2, return_alpha=True)
    np.testing.assert_allclose(alpha, g["gat_alpha"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(y, g["gat_y"], rtol=1e-5, atol=1e-6)
    # backward: gradient wrt x through BOTH the aggregate and the logits (el/er are functions of x)
    att = g["gat_att"]
    C = x.shape[2]
    gel, ger, gx = oracle.gat_bwd(ei, el, er, x, g["gat_g"], 0.2)
    gx_total = gx + gel[:, :, None] * att[:, :, :C] + ger[:, :, None] * att[:, :, C:]
    np.testing.assert_allclose(gx_total, g["gat_gx"], rtol=2e-4, atol=2e-5)


def test_gat_headmean_layer_fixture(golden, oracle):
    """gat_conv.py:98-122 with concat=False (reduce_mean over heads :115-118, + bias): the C oracle's GAT forward / backward
    composed with the dense parts in numpy, against the vector the reference's own segment ops produced (gatm_*)."""
    g = golden["layers"]
    ei, x, W, att, b = g["gat_ei"], g["gatm_x"], g["gatm_W"], g["gatm_att"], g["gatm_b"]
    H, C = att.shape[1], att.shape[2] // 2
    z = (x @ W).reshape(-1, H, C)
    el, er = (z * att[:, :, :C]).sum(-1), (z * att[:, :, C:]).sum(-1)
    y = oracle.gat_fwd(ei, el, er, z, 0.2).mean(axis=1) + b
    np.testing.assert_allclose(y, g["gatm_y"], rtol=1e-5, atol=1e-6)
    go = np.repeat(g["gatm_g"][:, None, :] / np.float32(H), H, axis=1).astype(np.float32)     # d mean / d head
    gel, ger, gz = oracle.gat_bwd(ei, el, er, z, go, 0.2)
    gz = gz + gel[:, :, None] * att[:, :, :C] + ger[:, :, None] * att[:, :, C:]
    np.testing.assert_allclose(x.T @ gz.reshape(x.shape[0], -1), g["gatm_gW"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gz.reshape(x.shape[0], -1) @ W.T, g["gatm_gx"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(g["gatm_g"].sum(0), g["gatm_gb"], rtol=1e-5, atol=1e-6)
    gatt = np.concatenate([(gel[:, :, None] * z).sum(0), (ger[:, :, None] * z).sum(0)], axis=-1)[None]
    np.testing.assert_allclose(gatt, g["gatm_gatt"], rtol=2e-4, atol=2e-5)


def test_sampler_restatement_against_reference_sample_adj(golden, oracle):
    """oracle.sample_adj_full (the Python restatement the sampler tests check against) vs the reference's own
    c_sample_adj on its deterministic branches (tests/golden/sampler.npz)."""
    import parity_cases as pc

    g = golden["sampler"]
    assert int(g["ncases"]) >= 19
    for ci in range(int(g["ncases"])):
        k = f"c{ci}"
        got = oracle.sample_adj_full(g[k + "_rowptr"], g[k + "_col"], g[k + "_seeds"])
        pc.compare_block_with_reference(got, g, k, f"restated sampler case {ci}")


def test_convert_statements_against_reference_ind2ptr(golden):
    """the numpy statements parity_cases.check_convert holds ind2ptr / ptr2ind to, vs the reference's compiled
    c_ind2ptr (tests/golden/convert.npz)."""
    g = golden["convert"]
    for ci in range(int(g["ncases"])):
        ind, M, ptr = g[f"v{ci}_ind"], int(g[f"v{ci}_M"]), g[f"v{ci}_ptr"]
        stated = np.concatenate(([0], np.cumsum(np.bincount(ind, minlength=M), dtype=np.int64)))
        assert same(stated, ptr), ci
        assert same(np.repeat(np.arange(M, dtype=np.int64), np.diff(ptr)), ind), ci


def test_oracle_rejects_out_of_range_ids(oracle):
    x = np.ones((3, 2), np.float32)
    with pytest.raises(IndexError):
        oracle.segment_max(x, np.array([0, 5, 1]), 3)  # segment_max_cpu.cpp:50
    with pytest.raises(IndexError):
        oracle.segment_sum(x, np.array([0, -1, 1]), 3)


def test_empty_inputs(oracle):
    x = np.zeros((0, 4), np.float32)
    ids = np.zeros((0,), np.int64)
    assert (oracle.segment_sum(x, ids, 3) == 0).all()
    assert (oracle.segment_mean(x, ids, 3) == 0).all()
    mx, arg = oracle.segment_max(x, ids, 3)
    assert (mx == 0).all() and (arg == 0).all()  # segment_max_cpu.cpp:28-30: zeros, before the lowest() fill
