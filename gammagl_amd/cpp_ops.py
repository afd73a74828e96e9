"""``torch.ops.ggl.*`` — the seven reference operators registered with the dispatcher FROM C++
(gammagl_amd/csrc/torch/ggl_torch.cpp -> gammagl_amd/lib/libggl_torch.so, ``TORCH_LIBRARY(ggl, m)``).

Same schemas as the Python-registered ``torch.ops.gammagl_amd.*`` (gammagl_amd/torch_ops.py) and the same kernels
behind them, but with no Python between the dispatcher and the C ABI: plan cache, launch policy and autograd
formulas are C++ (dispatcher -> libggl_torch.so -> include/ggl_mpops.h -> kernel).  That is the registration style the
reference's docs name as intended (register_cpp_ops.md:29-31); its code binds pybind11 functions instead
(src/operators.cpp:51-59).  ``gammagl_amd/compat/_torch_ext.py`` — the zero-edit drop-in for that pybind module —
binds these when the library is built.

    from gammagl_amd import cpp_ops
    ops = cpp_ops.load()                       # torch.ops.ggl
    out = ops.spmm_sum(edge_index, weight, x)  # CUDA tensors -> libggl_mpops_hip.so, CPU tensors -> libggl_mpops_host.so
"""
import os

import torch

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libggl_torch.so")
_loaded = False


def available():
    return os.path.exists(LIB_PATH)


def enabled():
    """Built, and not switched off with GGL_CPP_OPS=0 (A/B against the Python-registered ops)."""
    return available() and os.environ.get("GGL_CPP_OPS", "1") != "0"


def load():
    """Load libggl_torch.so (once) and return ``torch.ops.ggl``.  The kernel libraries themselves are opened by the
    C++ side on the first call for a device (a missing one raises there, naming the file)."""
    global _loaded
    if not _loaded:
        if not available():
            raise ImportError(f"{LIB_PATH} is not built: run `make -C gammagl_amd/csrc torch` "
                              f"(or __graft_entry__.build())")
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.ggl
