"""Drop-in for GammaGL's native module ``gammagl.mpops.torch_ext._torch_ext`` (the pybind11 module built from
gammagl/mpops/torch_ext/src/operators.cpp:51-59).

GammaGL binds its native message-passing ops in exactly one statement, ``gammagl/mpops/torch.py:3-7``::

    from .torch_ext._torch_ext import c_segment_sum, c_segment_mean, c_segment_max, \
        c_spmm_sum, c_spmm_mean, c_spmm_max, c_bspmm_sum

Copy (or symlink) THIS file to ``gammagl/mpops/torch_ext/_torch_ext.py`` — ``torch_ext`` has no ``__init__.py`` in
the reference tree, it is a namespace package, so a plain module of that name satisfies the import — and that
statement binds the seven callables below with ZERO edits to GammaGL: ``use_ext`` becomes True and every wrapper in
``mpops/torch.py`` (``unsorted_segment_*``, ``gspmm``, ``bspmm``), hence ``MessagePassing.propagate()`` and every
conv layer, runs on the MI355X kernels.  (If a compiled ``_torch_ext.*.so`` sits in the same directory, remove it: an
extension module wins over a ``.py`` of the same name.)

Same signatures and semantics as the pybind functions — ``(Tensor x, Tensor index, int64 N) -> Tensor`` for the
segment ops, ``(Tensor index, Tensor weight, Tensor x) -> Tensor`` for the SpMMs, autograd attached — dispatched on the
tensors' device exactly like the reference's ``x.is_cuda()`` / ``x.is_cpu()`` switch (src/segment_sum.cpp:19-33):
GPU tensors -> hand-written HIP for gfx950 (libggl_mpops_hip.so), CPU tensors -> the host build of the same kernel
sources (libggl_mpops_host.so).  Needs ``gammagl_amd`` importable (on ``sys.path`` / installed).

The callables bind the C++-registered dispatcher ops ``torch.ops.ggl.*`` (gammagl_amd/lib/libggl_torch.so: dispatcher ->
C++ -> C ABI -> kernel, no Python in between — like the pybind module they replace) when that library is built, else the
Python-registered ``torch.ops.gammagl_amd.*`` over the same kernels (bit-identical results, tests/test_torch_cpp.py).
"""
import torch

from gammagl_amd import cpp_ops as _cpp_ops

if _cpp_ops.enabled():
    _ops = _cpp_ops.load()
else:
    from gammagl_amd import torch_ops as _torch_ops   # registers torch.ops.gammagl_amd.* for the CUDA (HIP) and CPU keys

    _ops = _torch_ops.ops

__all__ = ["c_segment_sum", "c_segment_mean", "c_segment_max", "c_spmm_sum", "c_spmm_mean", "c_spmm_max", "c_bspmm_sum"]


def _index(index):
    # the pybind functions take the index as it comes and read it with data_ptr<int64_t>() (segment_sum_cpu.cpp:36):
    # "expected scalar type Long" for anything else — raised by the engine, same as there
    return index


def c_segment_sum(x, index, N):
    """SegmentSum::apply (src/segment_sum.cpp:35-54)."""
    return _ops.segment_sum(x, _index(index), int(N))


def c_segment_mean(x, index, N):
    """SegmentMean::apply (src/segment_mean.cpp:36-63)."""
    return _ops.segment_mean(x, _index(index), int(N))


def c_segment_max(x, index, N):
    """SegmentMax::apply (src/segment_max.cpp:37-61): returns the maxima (the argmax stays inside autograd)."""
    return _ops.segment_max(x, _index(index), int(N))[0]


def c_spmm_sum(index, weight, x):
    """SpMMSum::apply (src/gspmm.cpp:26-80)."""
    return _ops.spmm_sum(index, weight, x)


def c_spmm_mean(index, weight, x):
    """SpMMMean::apply (src/gspmm.cpp:82-141)."""
    return _ops.spmm_mean(index, weight, x)


def c_spmm_max(index, weight, x):
    """SpMMMax::apply (src/gspmm.cpp:143-202)."""
    return _ops.spmm_max(index, weight, x)


def c_bspmm_sum(index, weight, x):
    """BSpMMSum::apply (src/gspmm.cpp:204-260); returns a weight gradient like the reference does."""
    if isinstance(weight, torch.Tensor) and weight.dim() == 1 and x.dim() == 3 and x.shape[1] == 1:
        weight = weight.unsqueeze(1)      # mpops/torch.py:355-356 hands over a 1-D ones([E]) when weight is None
    return _ops.bspmm_sum(index, weight, x)
