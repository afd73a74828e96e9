"""Drop-in for GammaGL's GPU sparse-op module ``gammagl.ops.sparse._sparse_cuda`` (the pybind11 module built from
gammagl/ops/sparse/cuda/sparse_module.cu:10-16).

``gammagl/ops/sparse/sparse.py:26-29`` binds it in one statement::

    from ._sparse_cuda import (cuda_torch_ind2ptr, cuda_torch_ptr2ind, cuda_torch_neighbor_sample, cuda_torch_sample_adj)

and routes GPU tensors to these four names (CPU tensors keep going to GammaGL's own CPU extension,
``sparse.py:42-52,65-75,148-160``).  Copy (or symlink) THIS file to ``gammagl/ops/sparse/_sparse_cuda.py`` and that
statement binds the callables below with zero edits to GammaGL: ``ind2ptr`` / ``ptr2ind`` / ``sample_adj`` / ``neighbor_sample`` — hence
``SparseGraph.sample_adj`` and ``loader.NeighborSampler`` — run on the MI355X kernels (``ggl_ind2ptr``,
``ggl_ptr2ind``, ``ggl_sample_count`` / ``ggl_sample_pick``; include/ggl_mpops.h).  Same argument lists and return
values as the pybind functions (cuda/convert.cu:108-133, cuda/neighbor_sample.cu:882-929).  Needs ``gammagl_amd``
importable.
"""
import torch

from gammagl_amd import sampler as _sampler
from gammagl_amd import sparse as _sparse

__all__ = ["cuda_torch_ind2ptr", "cuda_torch_ptr2ind", "cuda_torch_neighbor_sample", "cuda_torch_sample_adj"]


def cuda_torch_ind2ptr(ind, M):
    """torch_cuda_ind2ptr (cuda/convert.cu:108-118): ptr[M + 1] from the sorted row indices."""
    return _sparse.ind2ptr(ind, int(M))


def cuda_torch_ptr2ind(ptr, E):
    """torch_cuda_ptr2ind (cuda/convert.cu:120-130)."""
    return _sparse.ptr2ind(ptr, int(E))


def _fanout(fanouts):
    f = fanouts.reshape(-1).tolist() if isinstance(fanouts, torch.Tensor) else list(fanouts)
    if len(f) != 1:
        raise RuntimeError(f"sample_adj samples ONE hop: got {len(f)} fan-outs")
    return int(f[0])


def cuda_torch_sample_adj(colptr, row, input_nodes, fanouts, replace=False, directed=False, random_seed=0):
    """torch_cu_sample_adj (cuda/neighbor_sample.cu:882-925): one hop; returns ``[rowptr, col, n_id, e_id]`` exactly as
    ``c_sample_adj`` does (sparse.py:157-166).  ``fanouts`` is the one-element CPU tensor sparse.py:159 builds;
    ``directed`` is accepted and unused as in the reference kernel; the draws come from the engine's device RNG stream
    (``Engine.reseed``), not from ``random_seed`` — the reference's are not reproducible either (sample.cpp: srand(time))."""
    out = _sampler.sample_adj(colptr, row, input_nodes, _fanout(fanouts), bool(replace))
    return list(out)


def cuda_torch_neighbor_sample(colptr, row, input_nodes, fanouts, replace=False, directed=False, random_seed=0):
    """torch_cu_neighbor_sample (cuda/neighbor_sample.cu:744-778), the multi-hop sampler: returns
    ``[sample_cols, sample_rows, sample_nodes, sample_edges]`` as cu_neighbor_sample assembles them (:704-733) — see
    ``gammagl_amd.sampler.neighbor_sample`` for the hop-by-hop statement.  ``fanouts`` is the CPU int64 tensor the
    caller builds (its data pointer is read on the host, :753); ``directed`` and ``random_seed`` are accepted: the
    reference kernel ignores the first and the draws here come from the engine's device RNG stream."""
    return _sampler.neighbor_sample(colptr, row, input_nodes, fanouts, bool(replace))
