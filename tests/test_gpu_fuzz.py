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
