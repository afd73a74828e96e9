"""``dgNN.operators.GATConvFuse`` on the MI355X kernels — the zero-edit drop-in for the one statement GammaGL's fused
GAT layer binds its kernel with (gammagl/layers/conv/fusedgat_conv.py:70-71)::

    from dgNN.operators import GATConvFuse
    self.op = GATConvFuse

and the one place it calls it (fusedgat_conv.py:121)::

    x = self.op(alpha_dst, alpha_src, row_ptr, col_ind, col_ptr, row_ind, permute, self.negative_slope, x, self.dropout_rate)

dgNN (github.com/dgSPARSE/dgNN, cited at fusedgat_conv.py:19-21; no version pinned anywhere in GammaGL, not vendored,
not installable here — SURVEY.md §8c) documents the op as ``GATConvFuse(attn_row, attn_col, row_ptr, col_ind, col_ptr,
row_ind, permute, negative_slope, in_feat, attn_drop)``:

    out[i, h, :] = sum over the entries (i, j) of CSR row i of
                   dropout(softmax_i(LeakyReLU(attn_row[i, h] + attn_col[j, h]))) * in_feat[j, h, :]

with ``row_ptr`` / ``col_ind`` the CSR of the AGGREGATING rows, ``col_ptr`` / ``row_ind`` its transpose and ``permute``
the CSR position of every CSC entry (all int32 in the layer, fusedgat_conv.py:113-117).  The layer builds that CSR on
``edge_index[0]`` (fusedgat_conv.py:106-108), so through this op it aggregates into ``edge_index[0]`` — the direction is
the caller's, kept as it is here (it coincides with GATConv's on symmetric graphs).

Here: the five tensors become a plan WITHOUT a sort (``Engine.graph_plan_from_csr``, cached on the identity + version of
``row_ptr``), and the op is ONE fused kernel per direction (csrc/gat.hip, gat_fast.hip): logits, edge softmax, attention
dropout and the weighted aggregate in the forward, both walks in the backward.  Autograd is attached (``attn_row``,
``attn_col``, ``in_feat`` get gradients; the structure tensors do not).  ``attn_drop`` is applied as passed — like
dgNN's op, which has no notion of train / eval: the layer passes ``self.dropout_rate`` in both.

Parity: dgNN itself is absent, so this boundary is pinned to the in-tree GATConv math (gat_conv.py:103-112 +
softmax.py:29-35) that the reference's own GATConv computes on the same graph: tests/test_cpu_backend.py (through a
stand-in package holding exactly the layer's import), tests/test_gpu_parity.py (the GAT goldens).
"""
import torch

from gammagl_amd import cpp_ops as _cpp_ops
from gammagl_amd import engine as _engine

# dispatcher -> C++ -> C ABI -> kernel (torch.ops.ggl.gat_fused_csr, csrc/torch/ggl_torch.cpp) when libggl_torch.so is
# built; else the ctypes engine over the same kernels (bit-identical, tests/test_torch_cpp.py)
_cpp = _cpp_ops.load() if _cpp_ops.enabled() else None

__all__ = ["GATConvFuse"]


def GATConvFuse(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute, negative_slope, in_feat, attn_drop=0.0):
    """Fused GAT aggregation over a caller-built CSR (see the module docstring).  Shapes: ``attn_row`` / ``attn_col``
    [N, H], ``in_feat`` [N, H, C] f32; ``row_ptr`` [N + 1], ``col_ind`` [E], ``col_ptr`` [N + 1], ``row_ind`` [E],
    ``permute`` [E] int32 or int64.  Returns [N, H, C]."""
    if not (isinstance(in_feat, torch.Tensor) and in_feat.dim() == 3):
        raise RuntimeError("GATConvFuse: in_feat must be [num_nodes, heads, channels]")
    if _cpp is not None:
        return _cpp.gat_fused_csr(row_ptr, col_ind, col_ptr, row_ind, permute, attn_col, attn_row, in_feat,
                                  float(negative_slope), float(attn_drop))
    eng = _engine(in_feat)
    n_rows, n_cols = int(row_ptr.shape[0]) - 1, int(col_ptr.shape[0]) - 1
    if attn_row.shape[0] != n_rows or attn_col.shape[0] != n_cols or in_feat.shape[0] != n_cols:
        raise RuntimeError(f"GATConvFuse: attn_row has {attn_row.shape[0]} rows for a CSR of {n_rows}, attn_col / in_feat "
                           f"{attn_col.shape[0]} / {in_feat.shape[0]} for {n_cols} columns")
    gp = eng.graph_plan_from_csr(row_ptr, col_ind, col_ptr, row_ind, permute, n_rows, n_cols)
    # engine naming: el = the term of the node gathered FROM (a CSR column), er = the term of the aggregating row
    return eng.gat_fused(gp, attn_col, attn_row, in_feat, float(negative_slope), num_nodes=n_rows,
                         dropout_rate=float(attn_drop), training=True)
