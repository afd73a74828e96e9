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
        # quarter-integers with |sum| < 2^11: every partial sum is exact in f32/f64/int32, so chunking
        # cannot change the result either; 16-bit floats round per add, so only unsplit rows are exact
        split = np.bincount(ids, minlength=N).max(initial=0) > chunk
        if dt in ("float32", "float64", "int32") or not split:
            pc.assert_same(got_s, ref_s, f"sum {prob} {dt}")
            pc.assert_same(got_m, ref_m, f"mean {prob} {dt}")
    finally:
        eng.chunk = old
        eng.seg_cache.clear()


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(problems())
def test_gspmm_fuzz(oracle, prob):
    run_gspmm_case(engine(), DEV, oracle, prob)


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
        np.testing.assert_allclose(pc.to_np(y), oy, rtol=2e-5, atol=2e-6)
        y.backward(pc.to_t(go, DEV))
        gel, ger, gx = oracle.gat_bwd(index, el, er, x, go, 0.2)
        np.testing.assert_allclose(pc.to_np(xt.grad), gx, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(pc.to_np(elt.grad), gel, rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(pc.to_np(ert.grad), ger, rtol=2e-4, atol=5e-5)
    finally:
        eng.chunk = old
        eng.seg_cache.clear(); eng.graph_cache.clear()
