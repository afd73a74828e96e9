"""Property-based fuzz of the kernel logic (host-emulated build, hypothesis): random sizes, feature
widths, dtypes, id distributions (sorted / unsorted / heavy rows / empty rows) and long-row thresholds,
every result compared with the oracle — bit-exact except float sums over chunked rows."""
import os
import subprocess

import numpy as np
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = torch.device("cpu")
_state = {}


def engine():
    if "eng" not in _state:
        subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
        from gammagl_amd import _lib
        from gammagl_amd.ops import Engine

        _state["eng"] = Engine(_lib.bind(os.path.join(HERE, "emul", "libggl_emul.so")), require_cuda=False)
    return _state["eng"]


@st.composite
def problems(draw):
    N = draw(st.integers(1, 40))
    E = draw(st.integers(0, 300))
    K = draw(st.sampled_from([1, 2, 3, 4, 5, 8, 12, 16, 47, 64, 65, 128, 256, 260]))
    chunk = draw(st.sampled_from([1, 2, 7, 64, 4096]))
    kind = draw(st.sampled_from(["uniform", "sorted", "hub", "single"]))
    seed = draw(st.integers(0, 2**31 - 1))
    return N, E, K, chunk, kind, seed


def make_ids(rng, N, E, kind):
    if kind == "single":
        return np.full(E, rng.integers(0, N), dtype=np.int64)
    ids = rng.integers(0, N, size=E).astype(np.int64)
    if kind == "hub" and E > 4:
        ids[: E // 2] = ids[0]
    if kind == "sorted":
        ids.sort()
    return ids


@settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck))
@given(problems(), st.sampled_from(["float32", "float64", "int32", "float16", "bfloat16"]))
def test_segment_ops_fuzz(oracle, prob, dt):
    run_segment_case(engine(), DEV, oracle, prob, dt)


def run_segment_case(eng, DEV, oracle, prob, dt):
    N, E, K, chunk, kind, seed = prob
    rng = np.random.default_rng(seed)
    ids = make_ids(rng, N, E, kind)
    vals = (rng.integers(-8, 9, size=(E, K)) * 0.25)
    bf = dt == "bfloat16"
    x = oracle.f32_to_bf16_bits(vals.astype(np.float32)) if bf else vals.astype(dt)
    old = eng.chunk
    eng.chunk = chunk
    eng.seg_cache.clear()
    try:
        xt, it = pc.to_t(x, DEV, dt), pc.to_t(ids, DEV)
        mx, arg = eng.segment_max_with_arg(xt, it, N)
        omx, oarg = oracle.segment_max(x, ids, N, bf16=bf)
        pc.assert_same(pc.to_np(mx), omx, f"max {prob} {dt}")
        pc.assert_same(pc.to_np(arg), oarg, f"argmax {prob} {dt}")
        got_s, got_m = pc.to_np(eng.c_segment_sum(xt, it, N)), pc.to_np(eng.c_segment_mean(xt, it, N))
        ref_s, ref_m = oracle.segment_sum(x, ids, N, bf16=bf), oracle.segment_mean(x, ids, N, bf16=bf)
        # quarter-integers with |sum| < 2^11: every partial sum is exact in f32/f64/int32, so chunking cannot change
        # the result; 16-bit floats round per add, and their rows are never chunked (hub rows: the row walk, or the
        # LDS-pipelined hub kernel on the GPU) — exact everywhere
        pc.assert_same(got_s, ref_s, f"sum {prob} {dt}")
        pc.assert_same(got_m, ref_m, f"mean {prob} {dt}")
    finally:
        eng.chunk = old
        eng.seg_cache.clear()


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(problems())
def test_gspmm_fuzz(oracle, prob):
    run_gspmm_case(engine(), DEV, oracle, prob)


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(problems(), st.sampled_from([0, 1]))
def test_gspmm_max_backward_mask_fuzz(oracle, prob, scatter):
    """the same cases with the max backward forced through the 1-bit winner mask (round 5), records in forward order / scattered"""
    eng = engine()
    with pc.option(eng, "maxbwd_mask", 1), pc.option(eng, "maxbwd_mask_kmax", 0), pc.option(eng, "maxbwd_mask_scatter", scatter):
        run_gspmm_case(eng, DEV, oracle, prob)


def run_gspmm_case(eng, DEV, oracle, prob):
    N, E, K, chunk, kind, seed = prob
    rng = np.random.default_rng(seed)
    index = np.stack([rng.integers(0, N, size=E), make_ids(rng, N, E, kind)]).astype(np.int64)
    w = (rng.integers(-4, 5, size=E) * 0.5).astype(np.float32)
    x = (rng.integers(-8, 9, size=(N, K)) * 0.25).astype(np.float32)
    go = (rng.integers(-8, 9, size=(N, K)) * 0.25).astype(np.float32)
    old = eng.chunk
    eng.chunk = chunk
    eng.seg_cache.clear(); eng.graph_cache.clear(); eng.w_cache.clear()
    old_deg = int(eng.lib.ggl_get_option(b"col_block_min_degree"))
    eng.set_option("col_block_min_degree", 0)   # wide cases take the column-block launches whatever the degree
    try:
        it, wt = pc.to_t(index, DEV), pc.to_t(w, DEV)
        for red, fn in (("sum", eng.c_spmm_sum), ("mean", eng.c_spmm_mean), ("max", eng.c_spmm_max)):
            xt = pc.to_t(x, DEV).requires_grad_(True)
            y = fn(it, wt, xt)
            if red == "sum":
                oy, ogx = oracle.spmm_sum_fwd(index, w, x), oracle.spmm_sum_bwd(index, w, go)
            elif red == "mean":
                oy, cnt = oracle.spmm_mean_fwd(index, w, x)
                ogx = oracle.spmm_mean_bwd(index, w, go, cnt)
            else:
                oy, a = oracle.spmm_max_fwd(index, w, x)
                ogx = oracle.spmm_max_bwd(index, w, go, a)
            if red == "mean":  # a division: exact only where the row was reduced in one piece
                np.testing.assert_allclose(pc.to_np(y), oy, rtol=1e-6, atol=1e-6)
            else:
                pc.assert_same(pc.to_np(y), oy, f"spmm {red} {prob}")  # exact products and sums
            if E > 0:
                y.backward(pc.to_t(go, DEV))
                np.testing.assert_allclose(pc.to_np(xt.grad), ogx, rtol=1e-6, atol=1e-6)
    finally:
        eng.chunk = old
        eng.set_option("col_block_min_degree", old_deg)
        eng.seg_cache.clear(); eng.graph_cache.clear(); eng.w_cache.clear()


@st.composite
def gat_problems(draw):
    N = draw(st.integers(1, 30))
    E = draw(st.integers(0, 250))
    H = draw(st.sampled_from([1, 2, 3, 4, 8]))
    C = draw(st.sampled_from([1, 3, 4, 8, 16, 64]))
    chunk = draw(st.sampled_from([1, 3, 16, 4096]))
    kind = draw(st.sampled_from(["uniform", "sorted", "hub", "single"]))
    scale = draw(st.sampled_from([0.1, 1.0, 20.0]))   # large logits: the online max moves a lot
    seed = draw(st.integers(0, 2**31 - 1))
    return N, E, H, C, chunk, kind, scale, seed


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(gat_problems())
def test_gat_fused_fuzz(oracle, prob):
    """One-walk online-softmax forward + two-walk backward vs the oracle's three-pass restatement of
    gat_conv.py:103-112: short rows, chunked hubs on either side, empty rows, big logits."""
    run_gat_case(engine(), DEV, oracle, prob)


def run_gat_case(eng, DEV, oracle, prob):
    N, E, H, C, chunk, kind, scale, seed = prob
    rng = np.random.default_rng(seed)
    index = np.stack([make_ids(rng, N, E, "hub" if kind == "single" else "uniform"),
                      make_ids(rng, N, E, kind)]).astype(np.int64)
    el = (rng.standard_normal((N, H)) * scale).astype(np.float32)
    er = (rng.standard_normal((N, H)) * scale).astype(np.float32)
    x = rng.standard_normal((N, H, C)).astype(np.float32)
    go = rng.standard_normal((N, H, C)).astype(np.float32)
    old = eng.chunk
    eng.chunk = chunk
    eng.seg_cache.clear(); eng.graph_cache.clear()
    try:
        elt, ert, xt = (pc.to_t(a, DEV).requires_grad_(True) for a in (el, er, x))
        y = eng.gat_fused(pc.to_t(index, DEV), elt, ert, xt, 0.2)
        oy = oracle.gat_fwd(index, el, er, x, 0.2)
        # (an output that cancels to ~0 carries the rounding of its O(|x|) terms)
        np.testing.assert_allclose(pc.to_np(y), oy, rtol=2e-5, atol=2e-6 * (1.0 + float(np.abs(x).max() if x.size else 0)))
        y.backward(pc.to_t(go, DEV))
        gel, ger, gx = oracle.gat_bwd(index, el, er, x, go, 0.2)
        # one ulp of a logit of magnitude L is 1.2e-7 L in the exponent of its softmax term: the tolerance follows the
        # largest logit (scale 8 draws logits of +-40), and the logit gradients — sums of terms that cancel — get an
        # absolute floor of the size of one term
        L = float(np.abs(el).max() + np.abs(er).max()) if el.size else 0.0
        rel = 2e-4 + 4 * 1.2e-7 * L
        term = float(np.abs(go).max() * np.abs(x).max()) if go.size else 0.0
        np.testing.assert_allclose(pc.to_np(xt.grad), gx, rtol=rel, atol=2e-5 * (1.0 + float(np.abs(go).max() if go.size else 0)))
        np.testing.assert_allclose(pc.to_np(elt.grad), gel, rtol=rel, atol=5e-5 + 4e-6 * C * term)
        np.testing.assert_allclose(pc.to_np(ert.grad), ger, rtol=rel, atol=5e-5 + 4e-6 * C * term)
    finally:
        eng.chunk = old
        eng.seg_cache.clear(); eng.graph_cache.clear()


@st.composite
def sampler_problems(draw):
    N = draw(st.integers(2, 40))
    E = draw(st.integers(0, 300))
    B = draw(st.integers(1, min(N, 12)))
    fanout = draw(st.sampled_from([-1, 1, 2, 5, 25]))
    replace = draw(st.booleans())
    seed = draw(st.integers(0, 2**31 - 1))
    return N, E, B, fanout, replace, seed


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(sampler_problems())
def test_sample_adj_fuzz(prob):
    """Structure of a sampled block (sample.cpp:10-135): row sizes, membership in the true neighbourhood,
    distinctness without replacement, seeds-first relabelling, columns ascending within a row, edge ids."""
    run_sampler_case(engine(), DEV, prob)


def run_sampler_case(eng, DEV, prob):
    from gammagl_amd.sampler import sample_adj

    N, E, B, fanout, replace, seed = prob
    rng = np.random.default_rng(seed)
    dst = np.sort(rng.integers(0, N, size=E)).astype(np.int64)     # CSR rows = destination
    col = rng.integers(0, N, size=E).astype(np.int64)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=N), out=rowptr[1:])
    seeds = rng.permutation(N)[:B].astype(np.int64)
    out_rowptr, local, n_id, e_pos = sample_adj(pc.to_t(rowptr, DEV), pc.to_t(col, DEV), pc.to_t(seeds, DEV), fanout,
                                                replace=replace, eng=eng)
    # the sort-free relabelling NeighborSampler uses (dense first-position scratch) gives the same block
    from gammagl_amd import sampler as _s
    scratch = torch.full((N,), _s._BIG, dtype=torch.int64, device=DEV)
    st0 = eng._rng_state(DEV).clone()
    out_rowptr, local, n_id, e_pos = (pc.to_np(t) for t in (out_rowptr, local, n_id, e_pos))
    eng._rng_state(DEV)[1] -= 1        # same draw again
    r2 = sample_adj(pc.to_t(rowptr, DEV), pc.to_t(col, DEV), pc.to_t(seeds, DEV), fanout, replace=replace, eng=eng,
                    first_pos=scratch)
    assert torch.equal(eng._rng_state(DEV), st0)
    for a_, b_ in zip((out_rowptr, local, n_id, e_pos), r2):
        np.testing.assert_array_equal(a_, pc.to_np(b_))
    assert bool((scratch == _s._BIG).all())
    assert list(n_id[:B]) == list(seeds) and len(set(n_id.tolist())) == len(n_id)
    assert out_rowptr[0] == 0 and out_rowptr[-1] == len(local) == len(e_pos)
    for i, s in enumerate(seeds):
        deg = rowptr[s + 1] - rowptr[s]
        want = deg if fanout < 0 else ((fanout if deg > 0 else 0) if replace else min(deg, fanout))
        lo, hi = out_rowptr[i], out_rowptr[i + 1]
        assert hi - lo == want, (i, deg, want)
        ep = e_pos[lo:hi]
        assert ((ep >= rowptr[s]) & (ep < rowptr[s + 1])).all()            # edges of this seed's CSR row
        assert (n_id[local[lo:hi]] == col[ep]).all()                       # local id <-> global neighbour
        if not replace:
            assert len(set(ep.tolist())) == len(ep)                        # distinct positions
        assert (np.diff(local[lo:hi]) >= 0).all()                          # columns ascending (sample.cpp:112-118)
    # every non-seed node of n_id is somebody's sampled neighbour
    assert set(n_id[B:].tolist()) <= set(col[e_pos].tolist())


@st.composite
def fused_problems(draw):
    N = draw(st.integers(1, 40))
    M = draw(st.integers(1, 40))
    E = draw(st.integers(0, 300))
    K = draw(st.sampled_from([4, 8, 12, 16, 64, 256, 260]))
    chunk = draw(st.sampled_from([1, 3, 64, 4096]))
    kind = draw(st.sampled_from(["uniform", "sorted", "hub", "single"]))
    relu = draw(st.booleans())
    p = draw(st.sampled_from([0.0, 0.0, 0.3, 0.7]))
    seed = draw(st.integers(0, 2**31 - 1))
    return N, M, E, K, chunk, kind, relu, p, seed


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(fused_problems())
def test_fused_epilogue_and_strided_fuzz(prob):
    run_fused_case(engine(), DEV, prob)


def run_fused_case(eng, DEV, prob):
    """(1) SpMM with the epilogue in its store == SpMM then the epilogue kernel on the same RNG state, bit for
    bit, values and gradients; (2) the strided / accumulating SpMM == the dense op on copies."""
    N, M, E, K, chunk, kind, relu, p, seed = prob
    rng = np.random.default_rng(seed)
    index = np.stack([rng.integers(0, M, size=E), make_ids(rng, N, E, kind)]).astype(np.int64)
    it = pc.to_t(index, DEV)
    g = torch.Generator().manual_seed(seed % 1000)
    w = torch.rand(E, generator=g).to(DEV)
    old = eng.chunk
    eng.chunk = chunk
    eng.seg_cache.clear(); eng.graph_cache.clear(); eng.w_cache.clear()
    try:
        gp = eng.graph_plan(it, N, M)
        xa = torch.randn(M, K, generator=g).to(DEV).requires_grad_(True)
        ba = torch.randn(1, K, generator=g).to(DEV).requires_grad_(True)
        xb, bb = xa.detach().clone().requires_grad_(True), ba.detach().clone().requires_grad_(True)
        go = torch.randn(N, K, generator=g).to(DEV)
        st_ = eng._rng_state(DEV).clone()
        ya = eng.spmm_bias_act(gp, w, xa, ba, relu=relu, p_drop=p, training=True)
        eng._rng_state(DEV).copy_(st_)
        yb = eng.bias_act(eng.spmm(gp, w, xb), bb, relu=relu, p_drop=p, training=True)
        assert torch.equal(ya, yb), prob
        ya.backward(go)
        yb.backward(go)
        assert torch.equal(xa.grad, xb.grad) and torch.equal(ba.grad, bb.grad), prob
        # strided / accumulating form on a column block of wider matrices
        KW = K + 8
        c0 = int(rng.integers(0, 3)) * 4
        xw = torch.randn(M, KW, generator=g).to(DEV)
        base = torch.randn(N, KW, generator=g).to(DEV)
        dense, _ = eng._spmm_fwd("sum", gp.fwd, gp.col, w, xw[:, c0:c0 + K].contiguous(), N)
        acc = base.clone()
        eng.spmm_sum_into(gp.fwd, gp.col, w, xw[:, c0:c0 + K], acc[:, c0:c0 + K], accumulate=True)
        np.testing.assert_allclose(pc.to_np(acc[:, c0:c0 + K]), pc.to_np(base[:, c0:c0 + K] + dense), rtol=1e-5, atol=1e-5)
        assert torch.equal(acc[:, :c0], base[:, :c0]) and torch.equal(acc[:, c0 + K:], base[:, c0 + K:])
        out = torch.full((N, KW), 3.0).to(DEV)
        eng.spmm_sum_into(gp.fwd, gp.col, w, xw[:, c0:c0 + K], out[:, c0:c0 + K])
        assert torch.equal(out[:, c0:c0 + K], dense)
    finally:
        eng.chunk = old
        eng.seg_cache.clear(); eng.graph_cache.clear(); eng.w_cache.clear()


# ---- bspmm: forward, input gradient, weight gradient (sorted-plan walk) -----------------------------------------------
@st.composite
def bspmm_problems(draw):
    N = draw(st.integers(1, 40))
    E = draw(st.integers(0, 600))
    H = draw(st.sampled_from([1, 2, 3, 8]))
    C = draw(st.sampled_from([4, 8, 16, 20, 32, 36, 44, 64, 100, 128, 256, 300]))
    kind = draw(st.sampled_from(["uniform", "sorted", "hub", "single"]))
    blocks = draw(st.booleans())
    seed = draw(st.integers(0, 2**31 - 1))
    return N, E, H, C, kind, blocks, seed


def run_bspmm_case(eng, DEV, oracle, prob):
    """bspmm_sum forward / gx / gw against the oracle bit for bit (quarter-integer data: every partial sum is exact, so
    neither the chunking of hub rows nor the 64-column-block launches of the weight-gradient walk can change a bit)."""
    N, E, H, C, kind, blocks, seed = prob
    rng = np.random.default_rng(seed)
    dst = make_ids(rng, N, E, kind)
    src = rng.integers(0, N, size=E).astype(np.int64)
    index = np.stack([src, dst])
    w = (rng.integers(-4, 5, size=(E, H)) * 0.5).astype(np.float32)
    x = (rng.integers(-8, 9, size=(N, H, C)) * 0.25).astype(np.float32)
    go = (rng.integers(-8, 9, size=(N, H, C)) * 0.25).astype(np.float32)
    names = (b"col_block_min_edges", b"col_block_min_degree")
    old = [eng.lib.ggl_get_option(n) for n in names]
    try:
        if blocks:
            for n in names:
                eng.set_option(n.decode(), 0)
        wt = pc.to_t(w, DEV).requires_grad_(True)
        xt = pc.to_t(x, DEV).requires_grad_(True)
        y = eng.c_bspmm_sum(pc.to_t(index, DEV), wt, xt)
        y.backward(pc.to_t(go, DEV))
        ogx, ogw = oracle.bspmm_sum_bwd(index, w, x, go)
        pc.assert_same(pc.to_np(y), oracle.bspmm_sum_fwd(index, w, x), f"bspmm y {prob}")
        pc.assert_same(pc.to_np(xt.grad), ogx, f"bspmm gx {prob}")
        pc.assert_same(pc.to_np(wt.grad), ogw, f"bspmm gw {prob}")
    finally:
        for n, v in zip(names, old):
            eng.set_option(n.decode(), v)
        eng.graph_cache.clear()
        eng.seg_cache.clear()


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(bspmm_problems())
def test_bspmm_fuzz(oracle, prob):
    run_bspmm_case(engine(), DEV, oracle, prob)
