"""``torch.ops.gammagl_amd.*`` — the HIP message-passing ops registered with the PyTorch dispatcher.

The reference binds its native ops as seven pybind11 free functions (src/operators.cpp:51-59) and notes
``TORCH_LIBRARY`` as the intended style (docs register_cpp_ops.md:29-31).  Here the same seven entry
points (+ the fused GAT op and the fused epilogue) are dispatcher ops:

    torch.ops.gammagl_amd.segment_sum(x, index, N)        -> Tensor          c_segment_sum
    torch.ops.gammagl_amd.segment_mean(x, index, N)       -> Tensor          c_segment_mean
    torch.ops.gammagl_amd.segment_max(x, index, N)        -> (Tensor, Tensor) c_segment_max (+ argmax)
    torch.ops.gammagl_amd.spmm_sum(index, weight, x)      -> Tensor          c_spmm_sum
    torch.ops.gammagl_amd.spmm_mean(index, weight, x)     -> Tensor          c_spmm_mean
    torch.ops.gammagl_amd.spmm_max(index, weight, x)      -> Tensor          c_spmm_max
    torch.ops.gammagl_amd.bspmm_sum(index, weight, x)     -> Tensor          c_bspmm_sum
    torch.ops.gammagl_amd.gat_fused(index, el, er, x, negative_slope, num_nodes, dropout_rate) -> Tensor
    torch.ops.gammagl_amd.bias_act(a, bias, relu, p_drop) -> Tensor

Kernels are registered for ``CUDA`` / ``AutogradCUDA`` (= HIP on ROCm: libggl_mpops_hip.so) and for ``CPU`` /
``AutogradCPU`` (libggl_mpops_host.so, the host build of the same kernel sources) — the reference's ops dispatch
on ``x.is_cuda()`` / ``x.is_cpu()`` the same way (src/segment_sum.cpp:19-33).  The dispatcher routes by the
tensors' device: a GPU tensor can only ever reach the HIP library (a missing one fails loudly at first use), a CPU
tensor only the host build.  Each kernel forwards to the engine entry point of the same name (gammagl_amd/ops.py),
i.e. ctypes -> C ABI (include/ggl_mpops.h) -> kernel; the autograd formulas are the ``torch.autograd.Function``s
defined there.  Fake (meta) kernels
give shapes/dtypes so the ops trace under ``torch.compile`` / FakeTensor without running.
"""
import torch
from torch.library import Library

NS = "gammagl_amd"

_SCHEMAS = {
    "segment_sum": "(Tensor x, Tensor index, int N) -> Tensor",
    "segment_mean": "(Tensor x, Tensor index, int N) -> Tensor",
    "segment_max": "(Tensor x, Tensor index, int N) -> (Tensor, Tensor)",
    "spmm_sum": "(Tensor index, Tensor? weight, Tensor x) -> Tensor",
    "spmm_mean": "(Tensor index, Tensor? weight, Tensor x) -> Tensor",
    "spmm_max": "(Tensor index, Tensor? weight, Tensor x) -> Tensor",
    "bspmm_sum": "(Tensor index, Tensor weight, Tensor x) -> Tensor",
    "gat_fused": "(Tensor index, Tensor el, Tensor er, Tensor x, float negative_slope=0.2, "
                 "int? num_nodes=None, float dropout_rate=0.0) -> Tensor",
    "bias_act": "(Tensor a, Tensor? bias, bool relu, float p_drop) -> Tensor",
}

_DEF = Library(NS, "DEF")
for _name, _sig in _SCHEMAS.items():
    _DEF.define(_name + _sig)
_IMPLS = []  # Library handles must stay alive for their registrations to stay
_ENGINES = {}  # dispatch key -> [callable returning the engine its kernels run on]


def _kernels(get_engine):
    """name -> python kernel, each a thin call into the engine returned by ``get_engine()``."""
    return {
        "segment_sum": lambda x, index, N: get_engine().c_segment_sum(x, index, N),
        "segment_mean": lambda x, index, N: get_engine().c_segment_mean(x, index, N),
        "segment_max": lambda x, index, N: get_engine().segment_max_with_arg(x, index, N),
        "spmm_sum": lambda index, weight, x: get_engine().c_spmm_sum(index, weight, x),
        "spmm_mean": lambda index, weight, x: get_engine().c_spmm_mean(index, weight, x),
        "spmm_max": lambda index, weight, x: get_engine().c_spmm_max(index, weight, x),
        "bspmm_sum": lambda index, weight, x: get_engine().c_bspmm_sum(index, weight, x),
        "gat_fused": lambda index, el, er, x, negative_slope=0.2, num_nodes=None, dropout_rate=0.0:
            get_engine().gat_fused(index, el, er, x, negative_slope, num_nodes, dropout_rate, True),
        "bias_act": lambda a, bias, relu, p_drop: get_engine().bias_act(a, bias, relu, p_drop, True),
    }


def register_backend(get_engine, backend="CUDA"):
    """Bind every op to ``get_engine()`` for dispatch key ``backend`` and its autograd key.

    The package registers ``CUDA`` (the MI355X engine) and ``CPU`` (the host build) at import; a later
    registration for the same key replaces the earlier one's engine (the CPU test-suite binds ``CPU`` to its own
    -O1 / AddressSanitizer builds of the same sources).
    """
    slot = _ENGINES.get(backend)
    if slot is not None:        # already registered: swap the engine the registered kernels resolve to
        slot[0] = get_engine
        return None
    slot = _ENGINES[backend] = [get_engine]
    lib = Library(NS, "IMPL")
    for name, fn in _kernels(lambda: slot[0]()).items():
        lib.impl(name, fn, backend)
        lib.impl(name, fn, "Autograd" + backend)
    _IMPLS.append(lib)
    return lib


def _register_fakes():
    lib = Library(NS, "IMPL")

    def seg(x, index, N):
        return x.new_empty((N,) + tuple(x.shape[1:]))

    def seg_max(x, index, N):
        shape = (N,) + tuple(x.shape[1:])
        return x.new_empty(shape), x.new_empty(shape, dtype=torch.int64)

    def like_x(index, weight, x):
        return torch.empty_like(x)  # gspmm.cpp:16 — out = zeros_like(x): square

    def gat(index, el, er, x, negative_slope=0.2, num_nodes=None, dropout_rate=0.0):
        n = x.shape[0] if num_nodes is None else num_nodes
        return x.new_empty((n,) + tuple(x.shape[1:]))

    for name, fn in (("segment_sum", seg), ("segment_mean", seg), ("segment_max", seg_max),
                     ("spmm_sum", like_x), ("spmm_mean", like_x), ("spmm_max", like_x),
                     ("bspmm_sum", like_x), ("gat_fused", gat),
                     ("bias_act", lambda a, bias, relu, p_drop: torch.empty_like(a))):
        lib.impl(name, fn, "Meta")
    _IMPLS.append(lib)


def _product_engine():
    from . import engine  # raises ImportError when the HIP library is missing: loud, no fallback

    return engine()


def _host_engine():
    import gammagl_amd

    if gammagl_amd._engine is not None and not gammagl_amd._engine.require_cuda:
        return gammagl_amd._engine     # an injected any-device engine (tests)
    return gammagl_amd.host_engine()


_register_fakes()
register_backend(_product_engine, "CUDA")
register_backend(_host_engine, "CPU")

ops = getattr(torch.ops, NS)
