"""gammagl_amd — MI355X-native message-passing backend for GammaGL's torch backend.

Scope: the ``gammagl.mpops`` scatter/segment reductions, the ``gspmm``/``bspmm``
SpMM and the fused GAT edge-softmax + aggregate that back ``MessagePassing.propagate()``;
hand-written HIP for gfx950 behind the reference's own Python op surface.

    from gammagl_amd import mpops          # drop-in for gammagl/mpops/torch.py
    from gammagl_amd import engine         # plans, explicit-plan ops, fused GAT

There is no CPU path: the HIP library must be built (``make -C gammagl_amd/csrc``) and tensors must
live on the GPU.
"""
__version__ = "0.1.0"

_engine = None


def engine():
    """Process-wide Engine bound to libggl_mpops_hip.so (raises ImportError if it is not built)."""
    global _engine
    if _engine is None:
        from . import _lib
        from .ops import Engine

        _engine = Engine(_lib.hip_lib(), require_cuda=True)
    return _engine
