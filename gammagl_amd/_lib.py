"""ctypes binding of include/ggl_mpops.h.

GPU tensors are served by exactly one library: ``gammagl_amd/lib/libggl_mpops_hip.so`` (hand-written HIP for
gfx950, built in-tree by ``gammagl_amd/csrc/Makefile``).  If it is missing or does not load, importing
the ops fails loudly — nothing in this package computes a GPU tensor's result anywhere else.  CPU tensors (the
reference dispatches on ``x.is_cpu()`` as well) are served by ``libggl_mpops_host.so``, the host build of the
same kernel sources; it is loaded on the first CPU call only.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "lib", "libggl_mpops_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libggl_mpops_host.so")

GGL_OK, GGL_EINVAL, GGL_EINDEX, GGL_EDTYPE, GGL_EHIP, GGL_EWORKSPACE = 0, -1, -2, -3, -4, -5
ABI_VERSION = 9


class SegPlanC(ctypes.Structure):
    """struct ggl_segplan (include/ggl_mpops.h)."""

    _fields_ = [
        ("rowptr", c_void_p), ("perm", c_void_p), ("long_rows", c_void_p), ("chunk_ptr", c_void_p),
        ("n_long", c_int64), ("n_chunks", c_int64), ("chunk", c_int64), ("partial", c_void_p),
        ("N", c_int64), ("E", c_int64), ("row_order", c_void_p), ("xcd_run_rows", c_int64),
        ("long_order", c_void_p),
        ("max_len", c_int64),
    ]


_P = POINTER(SegPlanC)
_V = c_void_p

# name -> (restype, argtypes): every symbol include/ggl_mpops.h declares
SIGNATURES = {
    "ggl_abi_version": (c_int, []),
    "ggl_last_error": (c_char_p, []),
    "ggl_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "ggl_plan_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_plan_build": (c_int, [_V, c_int64, c_int64, _V, _V, _V, c_size_t, _V, POINTER(c_int32),
                               POINTER(c_int64)]),
    "ggl_plan_long_workspace_bytes": (c_size_t, [c_int64]),
    "ggl_plan_long_count": (c_int, [_V, c_int64, c_int64, _V, c_size_t, _V, POINTER(c_int64),
                                    POINTER(c_int64)]),
    "ggl_plan_long_fill": (c_int, [_V, c_int64, c_int64, c_int64, _V, _V, _V, c_size_t, _V]),
    "ggl_partial_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int]),
    "ggl_fill_i64": (c_int, [_V, c_int64, c_int64, _V]),
    "ggl_gather_i64_to_i32": (c_int, [_V, _V, c_int64, _V, _V]),
    "ggl_gather_rows_f32": (c_int, [_V, _V, c_int64, c_int64, _V, _V]),
    "ggl_gather_rows_f32_ex": (c_int, [_V, c_int64, _V, c_int64, c_int64, _V, c_int64, _V]),
    "ggl_ind2ptr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_ind2ptr": (c_int, [_V, c_int64, c_int64, _V, _V, c_size_t, _V]),
    "ggl_ptr2ind": (c_int, [_V, c_int64, c_int64, _V, _V]),
    "ggl_sort_edges_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_sort_edges": (c_int, [_V, _V, c_int64, c_int64, _V, _V, c_size_t, _V]),
    "ggl_segment_sum": (c_int, [c_int, _V, _P, c_int64, _V, _V]),
    "ggl_segment_mean": (c_int, [c_int, _V, _P, c_int64, _V, _V]),
    "ggl_segment_max": (c_int, [c_int, _V, _P, c_int64, _V, _V, c_int64, _V]),
    "ggl_spmm_col_blocks": (c_int64, [c_int64, c_int64, c_int64]),
    "ggl_spmm_col_blocks_plan": (c_int64, [c_void_p, c_int64]),
    "ggl_segment_hub16_supported": (c_int, [c_int, c_int64, _V, _V]),
    "ggl_segment_hub16": (c_int, [c_int, c_int, _V, _P, c_int64, _V, _V]),
    "ggl_segment_sum_bwd": (c_int, [c_int, _V, _V, c_int64, c_int64, _V, _V]),
    "ggl_segment_mean_bwd": (c_int, [c_int, _V, _V, _V, c_int64, c_int64, _V, _V]),
    "ggl_segment_max_bwd": (c_int, [c_int, _V, _V, c_int64, c_int64, c_int64, _V, _V]),
    "ggl_spmm_sum": (c_int, [_P, _V, _V, c_int, _V, c_int64, _V, _V]),
    "ggl_spmm_sum_ex": (c_int, [_P, _V, _V, c_int, _V, c_int64, c_int64, _V, c_int64, c_int, _V]),
    "ggl_segment_sum_ex": (c_int, [c_int, _V, c_int64, _P, c_int64, _V, c_int64, c_int, _V]),
    "ggl_spmm_sum_bias_act": (c_int, [_P, _V, _V, c_int, _V, c_int64, _V, c_int, c_float, _V, _V, _V]),
    "ggl_spmm_mean": (c_int, [_P, _V, _V, c_int, _V, c_int64, _V, _V]),
    "ggl_spmm_max": (c_int, [_P, _V, _V, c_int, _V, c_int64, _V, _V, _V]),
    "ggl_spmm_mean_bwd": (c_int, [_P, _V, _V, c_int, _V, _V, c_int64, _V, _V]),
    "ggl_spmm_max_bwd": (c_int, [_P, _V, _V, c_int, _V, _V, c_int64, _V, _V]),
    "ggl_spmm_max_bwd32": (c_int, [_P, _V, _V, c_int, _V, _V, c_int64, _V, _V]),
    "ggl_spmm_max_mask_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_spmm_max_mask": (c_int, [_P, _V, _V, _V, c_int64, _V, _V]),
    "ggl_spmm_max_mask_words": (c_int64, [c_int64, c_int]),
    "ggl_spmm_max_bwd_mask": (c_int, [_P, _V, _V, c_int, _V, _V, _V, c_int64, _V, _V]),
    "ggl_invert_perm": (c_int, [_V, c_int64, _V, _V]),
    "ggl_bspmm_sum": (c_int, [_P, _V, _V, c_int, _V, c_int64, c_int64, _V, _V]),
    "ggl_bspmm_grad_w": (c_int, [_V, _V, _V, c_int64, c_int64, c_int64, _V, _V]),
    "ggl_bspmm_grad_w_sorted_scratch_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "ggl_bspmm_grad_w_sorted": (c_int, [_P, _V, _V, _V, _V, c_int64, c_int64, _V, _V, _V]),
    "ggl_colsum_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_colsum_f32": (c_int, [_V, c_int64, c_int64, _V, _V, c_size_t, _V]),
    "ggl_bias_act_fwd": (c_int, [_V, _V, c_int64, c_int64, c_int, c_float, _V, _V, _V]),
    "ggl_bias_act_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_bias_act_bwd": (c_int, [_V, _V, c_int64, c_int64, c_int, c_float, _V, _V, _V, _V, c_size_t, _V]),
    "ggl_spmm_epi_ex": (c_int, [_P, _V, _V, c_int, _V, c_int64, c_int64, _V, c_int64, c_int, c_int, _V, c_int64,
                                _V, c_int, c_float, _V, c_int64, c_int64, c_int, _V]),
    "ggl_segment_epi": (c_int, [_V, _P, c_int64, c_int, _V, c_int64, _V, c_int, c_float, _V, _V, _V]),
    "ggl_gat_fused_fwd": (c_int, [_P, _V, _V, _V, _V, c_float, c_int64, c_int64, c_float, _V, _V, _V, _V, _V]),
    "ggl_gat_partial_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "ggl_gat_fused_bwd_dst": (c_int, [_P, _V, _V, _V, _V, _V, _V, _V, _V, _V, c_float, c_int64, c_int64,
                                      c_float, _V, _V, _V, _V, _V, _V]),
    "ggl_gat_fused_bwd_src": (c_int, [_P, _V, _V, _V, _V, _V, c_int64, c_int64, _V, _V, _V]),
    "ggl_gat_fast_supported": (c_int, [c_int64, c_int64]),
    "ggl_gat_fast_fwd": (c_int, [_P, _V, _V, _V, _V, c_int64, c_float, c_int64, c_int64, c_float, _V, _V, _V, _V, _V]),
    "ggl_gat_fast_bwd": (c_int, [_P, _V, _P, _V, _V, _V, _V, _V, _V, _V, _V, _V, c_float, c_int64, c_int64, c_float,
                                 _V, _V, _V, _V, _V, _V]),
    "ggl_gat_sh_supported": (c_int, [c_int64, c_int64, c_int64]),
    "ggl_gat_sh_partial_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_gat_sh_fwd": (c_int, [_P, _V, _V, _V, _V, c_int64, c_float, c_float, _V, _V, _V, _V, _V]),
    "ggl_gat_sh_stats": (c_int, [_V, _V, _V, _V, _V, c_int64, c_int64, _V, _V]),
    "ggl_gat_sh_bwd": (c_int, [_P, _V, _P, _V, _V, _V, _V, c_int64, _V, _V, _V, _V, c_int64, c_float, c_float, _V,
                               _V, _V, _V, _V]),
    "ggl_sample_count": (c_int, [_V, _V, c_int64, c_int64, c_int64, c_int, _V, _V]),
    "ggl_sample_pick": (c_int, [_V, _V, _V, c_int64, c_int64, c_int, _V, _V, _V, _V, _V]),
    "ggl_sample_hop_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_sample_hop": (c_int, [_V, _V, _V, _V, c_int64, c_int64, c_int64, c_int64, c_int64, _V, _V, _V, _V, _V, _V, _V,
                               _V, c_size_t, _V]),
    "ggl_sample_hop_ex": (c_int, [_V, _V, _V, _V, c_int64, c_int64, c_int64, c_int64, c_int64, _V, _V, _V, _V, _V, _V, _V,
                                  _V, c_size_t, _V, _V]),
    "ggl_block_transpose_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ggl_block_transpose": (c_int, [_V, _V, c_int64, c_int64, c_int64, _V, _V, _V, c_size_t, _V]),
    "ggl_set_option": (c_int, [c_char_p, c_int64]),
    "ggl_get_option": (c_int64, [c_char_p]),
    "ggl_policy_chunk": (c_int64, [c_int64]),
    "ggl_policy_spmm_width": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "ggl_policy_head_channels": (c_int64, [c_int64, c_int64, c_int64]),
    "ggl_policy_mean_bwd_prescale": (c_int, [c_int64, c_int64]),
    "ggl_policy_gradw_sorted": (c_int, [c_int64, c_int64]),
    "ggl_policy_maxbwd_form": (c_int, [c_int64, c_int64, c_int64]),
    "ggl_policy_xcd_run_rows": (c_int64, [c_int64, ctypes.c_double]),
    "ggl_policy_row_order": (c_int, [POINTER(c_int64), POINTER(c_int64)]),
    "ggl_calib_stream": (c_int, [_V, _V, c_int64, c_int, _V]),
    "ggl_time_spmm_sum": (c_int, [_P, _V, _V, c_int, _V, c_int64, _V, _V, c_int, POINTER(c_float)]),
}


def bind(path):
    """dlopen `path` and attach the prototypes of every symbol the header declares."""
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.ggl_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {got}, expected {ABI_VERSION}")
    return lib


_hip = None


def hip_lib():
    """The HIP library (singleton).  Raises ImportError when it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError(
                f"{HIP_LIB_PATH} not found: build it with `make -C gammagl_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()').  gammagl_amd has no "
                "CPU / PyTorch fallback.")
        _hip = bind(HIP_LIB_PATH)
    return _hip


_host = None


def host_lib():
    """The host build of the kernel sources (singleton): the CPU dispatch key's library."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} not found: build it with `make -C gammagl_amd/csrc host`")
        _host = bind(HOST_LIB_PATH)
    return _host
