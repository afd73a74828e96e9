"""The same property-based cases as tests/test_emul_fuzz.py, on the real MI355X through the product
library (derandomised, fewer examples): catches anything the host-emulated build cannot show — the
device's expf, rounding of contracted operations, scalar-path index loads, wave-level scheduling."""
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import test_emul_fuzz as F

pytestmark = pytest.mark.gpu
import os  # noqa: E402

# defaults: 100 derandomised examples per property; GGL_FUZZ_EXAMPLES / GGL_FUZZ_RANDOM=1 widen a manual hunt
_cfg = dict(max_examples=int(os.environ.get("GGL_FUZZ_EXAMPLES", "100")), deadline=None,
            derandomize=os.environ.get("GGL_FUZZ_RANDOM", "0") != "1", suppress_health_check=list(HealthCheck))


@pytest.fixture(scope="module")
def target():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; the HIP path has no fallback")
    from gammagl_amd import engine

    return engine(), torch.device("cuda", 0)


@settings(**_cfg)
@given(F.problems(), st.sampled_from(["float32", "float64", "int32", "float16", "bfloat16"]))
def test_segment_ops_fuzz_gpu(target, oracle, prob, dt):
    F.run_segment_case(target[0], target[1], oracle, prob, dt)


@settings(**_cfg)
@given(F.problems())
def test_gspmm_fuzz_gpu(target, oracle, prob):
    F.run_gspmm_case(target[0], target[1], oracle, prob)


@settings(**_cfg)
@given(F.problems())
def test_gspmm_max_backward_mask_fuzz_gpu(target, oracle, prob):
    """the same cases with the max backward forced through the 1-bit winner mask (round 5: v_writelane-assembled records)"""
    with F.pc.option(target[0], "maxbwd_mask", 1), F.pc.option(target[0], "maxbwd_mask_kmax", 0):
        F.run_gspmm_case(target[0], target[1], oracle, prob)


@settings(**_cfg)
@given(F.gat_problems())
def test_gat_fused_fuzz_gpu(target, oracle, prob):
    F.run_gat_case(target[0], target[1], oracle, prob)


@settings(**_cfg)
@given(F.sampler_problems())
def test_sample_adj_fuzz_gpu(target, prob):
    F.run_sampler_case(target[0], target[1], prob)


@settings(**_cfg)
@given(F.fused_problems())
def test_fused_epilogue_and_strided_fuzz_gpu(target, prob):
    F.run_fused_case(target[0], target[1], prob)


@settings(**_cfg)
@given(F.bspmm_problems())
def test_bspmm_fuzz_gpu(target, oracle, prob):
    """incl. the LDS-staged weight-gradient walk (edgedot.hip), which only the GPU build has"""
    F.run_bspmm_case(target[0], target[1], oracle, prob)


@st.composite
def _headmean_problems(draw):
    N = draw(st.integers(1, 40))
    E = draw(st.integers(0, 300))
    F_in = 4 * draw(st.integers(1, 16))
    C = draw(st.integers(1, 64))
    chunk = draw(st.sampled_from([1, 3, 16, 4096]))
    kind = draw(st.sampled_from(["uniform", "sorted", "hub", "single"]))
    scale = draw(st.sampled_from([0.05, 1.0, 8.0]))
    p_drop = draw(st.sampled_from([0.0, 0.0, 0.5]))
    seed = draw(st.integers(0, 2**31 - 1))
    return N, E, F_in, C, chunk, kind, scale, p_drop, seed


@settings(**_cfg)
@given(_headmean_problems())
def test_gat_headmean_fuzz_gpu(target, prob):
    """The aggregate-then-transform output layer (shared-row kernels: row broadcasts, 16-lane reduce-scatter, hub
    chunks, partial blocks at row ends) == the same layer on the transform-then-aggregate kernels, forward and the
    gradients of x / W / att, with the same dropout mask for the same rng state."""
    import numpy as np

    from gammagl_amd import layers

    eng, dev = target
    N, E, F_in, C, chunk, kind, scale, p_drop, seed = prob
    rng = np.random.default_rng(seed)
    index = np.stack([F.make_ids(rng, N, E, "hub" if kind == "single" else "uniform"),
                      F.make_ids(rng, N, E, kind)]).astype(np.int64)
    ei = torch.as_tensor(index, device=dev)
    x = torch.as_tensor(rng.standard_normal((N, F_in)).astype(np.float32), device=dev)
    go = torch.as_tensor(rng.standard_normal((N, C)).astype(np.float32), device=dev)
    torch.manual_seed(seed % 1000)
    conv = layers.FusedGATConv(F_in, C, heads=8, concat=False, dropout_rate=p_drop).to(dev)
    with torch.no_grad():
        conv.w.mul_(scale / 0.05)
        conv.att.mul_(scale / 0.05)
    conv.train()
    old = eng.chunk
    eng.chunk = chunk
    eng.clear_caches()
    try:
        outs = []
        for fast in (True, False):
            eng.gat_fast = fast
            eng.reseed(seed % 997)
            eng._rng_state(dev)
            for p_ in conv.parameters():
                p_.grad = None
            xa = x.clone().requires_grad_(True)
            y = conv(xa, ei, N)
            y.backward(go)
            outs.append([y.detach(), xa.grad, conv.w.grad.clone(), conv.att.grad.clone()])
        # absolute floor: gradients that cancel to ~0 (a single edge has no logit gradient) carry rounding noise of
        # the size of the terms that cancel
        terms = (1.0 + float(conv.w.abs().max()) * float(conv.att.abs().max())) * (1.0 + float(x.abs().max())) \
            * (1.0 + float(go.abs().max())) * (1.0 + float(conv.w.abs().max()))
        # the two paths subtract the row maximum from logits of different provenance (a max pre-pass vs the online
        # softmax): with |logit| ~ 10^3 (scale 8) one ulp of the logit is already 1e-4 in the exponent
        with torch.no_grad():
            xw = (x @ conv.w).reshape(N, 8, C)
            logit = float((xw * conv.att[:, :, :C]).sum(-1).abs().max() + (xw * conv.att[:, :, C:]).sum(-1).abs().max())
        rel = 3e-4 + 8 * 1.2e-7 * logit
        for a, b, nm in zip(outs[0], outs[1], ("y", "gx", "gW", "gatt")):
            tol = rel * float(b.abs().max()) + 1e-6 * terms
            assert float((a - b).abs().max()) <= tol, (prob, nm, float((a - b).abs().max()), tol)
        if p_drop == 0.0 and E > 0:
            # round 6: and against GROUND TRUTH — the layer in float64 (oracle/parity.py gat_conv_composed, torch scatters), with the
            # same composition in float32 (torch ops, none of this library's kernels) as the yardstick:
            # err(head-mean HIP) <= max(1e-5, 2 err(f32 composition)).  Scale floor of the gradients that can cancel to exactly 0
            # (one node with 42 parallel self-loops: every logit equal, every softmax gradient 0 in exact arithmetic): 1e-2 of the
            # bound `terms` on the cancelling terms' magnitude — the criterion then admits 1e-7 of those terms, f32's own noise
            from oracle import parity

            def composed(dtype):
                ps = [t.detach().to(dtype).requires_grad_(True) for t in (x, conv.w, conv.att, conv.bias)]
                out = parity.gat_conv_composed(*ps, ei, N, 8, C, concat=False, slope=conv.negative_slope)
                out.backward(go.to(dtype))
                return [out.detach(), ps[0].grad, ps[1].grad, ps[2].grad]

            truth, f32 = composed(torch.float64), composed(torch.float32)
            names = ("y", "gx", "gW", "gatt")
            e_hip = parity.layer_errors_vs_truth(truth, outs[0], names, zero_mean_rows=("gx", "gW", "gatt"), abs_floor=1e-2 * terms)
            e_f32 = parity.layer_errors_vs_truth(truth, f32, names, zero_mean_rows=("gx", "gW", "gatt"), abs_floor=1e-2 * terms)
            for nm in names:
                assert e_hip[nm] <= max(1e-5, 2.0 * e_f32[nm]), (prob, nm, e_hip, e_f32)
    finally:
        eng.gat_fast = True
        eng.chunk = old
        eng.clear_caches()
