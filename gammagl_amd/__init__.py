"""gammagl_amd — MI355X-native message-passing backend for GammaGL's torch backend.

Scope: the ``gammagl.mpops`` scatter/segment reductions, the ``gspmm``/``bspmm``
SpMM and the fused GAT edge-softmax + aggregate that back ``MessagePassing.propagate()``;
hand-written HIP for gfx950 behind the reference's own Python op surface.

    from gammagl_amd import mpops          # drop-in for gammagl/mpops/torch.py
    from gammagl_amd import engine         # plans, explicit-plan ops, fused GAT

GPU tensors run on ``lib/libggl_mpops_hip.so`` and on nothing else: if it is missing the import of the ops
fails loudly.  CPU tensors — the reference's ops dispatch on ``x.is_cpu()`` too (src/segment_sum.cpp:19-33;
BASELINE config 1 is ``--gpu -1``) — run on ``lib/libggl_mpops_host.so``, the host build of the same kernel
sources (csrc/host/host_shim.hpp).  A tensor is never moved between devices behind the caller's back.
"""
__version__ = "0.2.0"

_engine = None        # the MI355X engine (tests may inject an Engine built on another library here)
_host_engine = None   # the CPU backend (host build of the kernel sources), created on first CPU call


def engine(like=None):
    """The Engine for `like` (a tensor, a device, or None = the MI355X engine): GPU tensors -> the engine bound to
    libggl_mpops_hip.so (ImportError if it is not built), CPU tensors -> the host build."""
    global _engine
    if like is not None:
        dev = getattr(like, "device", like)
        if str(dev).startswith("cpu") and not (_engine is not None and not _engine.require_cuda):
            return host_engine()
    if _engine is None:
        from . import _lib
        from .ops import Engine

        _engine = Engine(_lib.hip_lib(), require_cuda=True)
        _engine.is_product = True     # bound to gammagl_amd/lib/libggl_mpops_hip.so: the library torch.ops.ggl serves GPU tensors from
    return _engine


def host_engine():
    """Process-wide Engine bound to libggl_mpops_host.so: CPU tensors only."""
    global _host_engine
    if _host_engine is None:
        from . import _lib
        from .ops import Engine

        _host_engine = Engine(_lib.host_lib(), require_cuda=False, cpu_only=True)
        _host_engine.is_product = True
    return _host_engine
