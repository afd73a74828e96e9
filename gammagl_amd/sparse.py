"""Device-side format conversion, the step right before every kernel:

* ``ind2ptr(ind, M)`` / ``ptr2ind(ptr, E)`` — ``gammagl/ops/sparse`` (C++ ``cpu/convert.cpp:58-128``,
  CUDA ``cuda/convert.cu:41-104``, numpy fallback ``ops/sparse/__init__.py:23-41``);
* ``sort_edge_index(edge_index, edge_attr, num_nodes, sort_by_row)`` — ``utils/sort_edge_index.py:5-44``.

``FusedGATConv.forward`` converts to numpy and back for these (``fusedgat_conv.py:106-117``); here they
stay on the GPU (rocPRIM radix sort + binary-search kernels in ``csrc/plan.hip``).  Results are the
reference's, with one strengthening: ties in ``sort_edge_index`` keep their original order (the
reference's argsort leaves them unspecified).
"""
import torch

from . import engine as _engine
from .ops import _ptr


def ind2ptr(ind, M, eng=None):
    """ptr[M+1] with ptr[r+1]-ptr[r] = number of entries of ``ind`` equal to r (int64)."""
    eng = eng or _engine(ind)      # (CPU tensors: the host build, as the reference's c_ind2ptr serves them)
    dev = eng._dev(ind)
    ind = ind.contiguous().to(torch.int64)
    E, M = int(ind.shape[0]), int(M)
    ptr = torch.empty(M + 1, dtype=torch.int64, device=dev)
    wsb = eng.lib.ggl_ind2ptr_workspace_bytes(E, M)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    eng._check(eng.lib.ggl_ind2ptr(_ptr(ind), E, M, _ptr(ptr), _ptr(ws), wsb, eng._stream(dev)))
    return ptr


def ptr2ind(ptr, E=None, eng=None):
    """ind[p] = r for ptr[r] <= p < ptr[r+1] (int64 [E]; E defaults to ptr[-1])."""
    eng = eng or _engine(ptr)
    dev = eng._dev(ptr)
    ptr = ptr.contiguous().to(torch.int64)
    M = int(ptr.shape[0]) - 1
    total = int(ptr[-1]) if M >= 0 else 0
    E = total if E is None else min(int(E), total)
    ind = torch.empty(E, dtype=torch.int64, device=dev)
    eng._check(eng.lib.ggl_ptr2ind(_ptr(ptr), M, E, _ptr(ind), eng._stream(dev)))
    return ind


def sort_edge_index(edge_index, edge_attr=None, num_nodes=None, sort_by_row=True, eng=None):
    """Row-wise (or column-wise) lexicographic sort of ``edge_index`` and its attributes."""
    eng = eng or _engine(edge_index)
    dev = eng._dev(edge_index)
    ei = edge_index.contiguous().to(torch.int64)
    E = int(ei.shape[1])
    if num_nodes is None:  # utils/num_nodes.py: max id + 1
        num_nodes = int(ei.max()) + 1 if E > 0 else 0
    major, minor = (ei[0], ei[1]) if sort_by_row else (ei[1], ei[0])
    perm = torch.empty(E, dtype=torch.int32, device=dev)
    wsb = eng.lib.ggl_sort_edges_workspace_bytes(E, int(num_nodes))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    eng._check(eng.lib.ggl_sort_edges(_ptr(major), _ptr(minor), E, int(num_nodes), _ptr(perm), _ptr(ws),
                                      wsb, eng._stream(dev)))
    p = perm.long()
    out = ei.index_select(1, p)
    if edge_attr is None:
        return out
    if torch.is_tensor(edge_attr):
        return out, edge_attr.index_select(0, p)
    return out, [e.index_select(0, p) for e in edge_attr]


__all__ = ["ind2ptr", "ptr2ind", "sort_edge_index"]
