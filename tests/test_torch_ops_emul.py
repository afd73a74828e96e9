"""torch.ops.gammagl_amd.* (gammagl_amd/torch_ops.py) exercised WITHOUT a GPU, through the ``CPU`` dispatch key:
first as the package registers it (libggl_mpops_host.so, the host build of the kernel sources), then with the key's
engine swapped for the tests' own -O1 build, checked against the oracle, for gradients and for the schema /
fake-tensor kernels with ``torch.library.opcheck``."""
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ops():
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    from gammagl_amd import _lib, torch_ops
    from gammagl_amd.ops import Engine

    eng = Engine(_lib.bind(os.path.join(HERE, "emul", "libggl_emul.so")), require_cuda=False)
    torch_ops.register_backend(lambda: eng, "CPU")       # the key is registered already: swaps its engine
    yield torch_ops.ops
    torch_ops.register_backend(torch_ops._host_engine, "CPU")


def _graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (2, e), generator=g), g


def test_product_registers_cpu_and_hip_keys(golden):
    """As shipped: CPU tensors dispatch to the host build of the kernel sources (the reference's ops dispatch on
    x.is_cpu() too: BASELINE config 1 runs with --gpu -1), GPU tensors to the HIP library only; the engines refuse the
    other device's tensors instead of moving them."""
    import gammagl_amd
    from gammagl_amd import _lib, torch_ops

    assert set(torch_ops._ENGINES) == {"CUDA", "CPU"}
    x = torch.ones(3, 2)
    ids = torch.tensor([0, 1, 1])
    assert torch_ops.ops.segment_sum(x, ids, 2).tolist() == [[1.0, 1.0], [2.0, 2.0]]
    host = gammagl_amd.host_engine()
    assert host.cpu_only and host.lib is _lib.host_lib() and gammagl_amd.engine(x) is host
    assert os.path.basename(_lib.HOST_LIB_PATH) == "libggl_mpops_host.so"
    # the reference's own known answers through the shipped CPU key, all three reductions
    g = golden["kat"]
    idx = torch.from_numpy(g["idx"].copy())
    for op, fn in (("sum", torch_ops.ops.segment_sum), ("mean", torch_ops.ops.segment_mean)):
        xk = torch.from_numpy(g[f"{op}_float32_d2_x"].copy())
        np.testing.assert_array_equal(fn(xk, idx, 2).numpy(), g[f"{op}_float32_d2_y"])
    xk = torch.from_numpy(g["max_float32_d2_x"].copy())
    np.testing.assert_array_equal(torch_ops.ops.segment_max(xk, idx, 2)[0].numpy(), g["max_float32_d2_y"])
    # every op of the reference's pybind module has a dispatcher schema
    for name in ("segment_sum", "segment_mean", "segment_max", "spmm_sum", "spmm_mean", "spmm_max",
                 "bspmm_sum", "gat_fused", "bias_act"):
        assert hasattr(torch_ops.ops, name)


def test_segment_ops_match_oracle(ops, oracle):
    ei, g = _graph(37, 300, 0)
    ids = ei[1]
    x = torch.randn(300, 12, generator=g)
    np.testing.assert_array_equal(ops.segment_sum(x, ids, 37).numpy(),
                                  oracle.segment_sum(x.numpy(), ids.numpy(), 37))
    np.testing.assert_array_equal(ops.segment_mean(x, ids, 37).numpy(),
                                  oracle.segment_mean(x.numpy(), ids.numpy(), 37))
    out, arg = ops.segment_max(x, ids, 37)
    ro, ra = oracle.segment_max(x.numpy(), ids.numpy(), 37)
    np.testing.assert_array_equal(out.numpy(), ro)
    np.testing.assert_array_equal(arg.numpy(), ra)
    assert arg.dtype == torch.int64 and not arg.requires_grad


def test_autograd_through_dispatcher(ops, oracle):
    ei, g = _graph(29, 200, 1)
    w = torch.rand(200, generator=g)
    x = torch.randn(29, 8, generator=g, requires_grad=True)
    out = ops.spmm_sum(ei, w, x)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    np.testing.assert_allclose(out.detach().numpy(), oracle.spmm_sum_fwd(ei.numpy(), w.numpy(), x.detach().numpy()),
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), oracle.spmm_sum_bwd(ei.numpy(), w.numpy(), gout.numpy()),
                               rtol=1e-6, atol=1e-6)
    # segment_max: gradient lands on the argmax rows only
    xs = torch.randn(200, 4, generator=g, requires_grad=True)
    o, arg = ops.segment_max(xs, ei[1], 29)
    o.sum().backward()
    expect = torch.zeros(201, 4).scatter_(0, arg.clamp(max=200), torch.ones_like(o))[:200]
    np.testing.assert_array_equal(xs.grad.numpy(), expect.numpy())
    # under no_grad / inference_mode the same kernels run without recording a graph
    with torch.inference_mode():
        assert torch.equal(ops.spmm_sum(ei, w, x.detach()), out.detach())


def test_bspmm_gat_bias_act(ops):
    ei, g = _graph(23, 150, 2)
    x = torch.randn(23, 4, 8, generator=g, requires_grad=True)
    wh = torch.rand(150, 4, generator=g, requires_grad=True)
    ref = torch.zeros(23, 4, 8).index_add_(0, ei[1], x.detach()[ei[0]] * wh.detach().unsqueeze(-1))
    out = ops.bspmm_sum(ei, wh, x)
    np.testing.assert_allclose(out.detach().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    out.sum().backward()
    assert x.grad is not None and wh.grad is not None  # the reference populates w.grad too (gspmm.cpp:259)
    el = torch.randn(23, 4, generator=g)
    er = torch.randn(23, 4, generator=g)
    o = ops.gat_fused(ei, el, er, x.detach())
    e = torch.nn.functional.leaky_relu(el[ei[0]] + er[ei[1]], 0.2)
    m = torch.full((23, 4), -3.4e38).scatter_reduce_(0, ei[1].view(-1, 1).expand_as(e), e, "amax")
    p = torch.exp(e - m[ei[1]])
    den = torch.zeros(23, 4).index_add_(0, ei[1], p)
    ref = torch.zeros(23, 4, 8).index_add_(0, ei[1], (p / (den[ei[1]] + 1e-16)).unsqueeze(-1) * x.detach()[ei[0]])
    np.testing.assert_allclose(o.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    a = torch.randn(23, 16, generator=g)
    b = torch.randn(16, generator=g)
    np.testing.assert_array_equal(ops.bias_act(a, b, True, 0.0).numpy(), torch.relu(a + b).numpy())


def test_opcheck_schema_and_fake(ops):
    ei, g = _graph(11, 60, 3)
    x = torch.randn(60, 5, generator=g)
    xn = torch.randn(11, 4, generator=g)
    w = torch.rand(60, generator=g)
    utils = ("test_schema", "test_faketensor")
    torch.library.opcheck(ops.segment_sum.default, (x, ei[1], 11), test_utils=utils)
    torch.library.opcheck(ops.segment_max.default, (x, ei[1], 11), test_utils=utils)
    torch.library.opcheck(ops.spmm_sum.default, (ei, w, xn), test_utils=utils)
    torch.library.opcheck(ops.spmm_mean.default, (ei, None, xn), test_utils=utils)
    torch.library.opcheck(ops.gat_fused.default, (ei, torch.randn(11, 2), torch.randn(11, 2),
                                                  torch.randn(11, 2, 3)), test_utils=utils)
