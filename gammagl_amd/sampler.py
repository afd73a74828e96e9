"""Neighbour sampling on the device: ``sample_adj`` and the ``NeighborSampler``
loop around it, as GammaGL's mini-batch GraphSAGE pipeline uses them
(``ops/sparse/cpu/sample.cpp:10-135``; ``loader/neighbor_sampler.py:29-112``;
``examples/graphsage/reddit_sage_trainer.py:55-57``).

Same outputs as the reference's ``sample_adj`` — ``(out_rowptr, out_col, out_n_id, out_e_id)`` with
``out_n_id`` = the seeds followed by newly met nodes in first-seen order and every row's columns sorted
by local id — but produced by two HIP kernels (count, pick: Floyd's algorithm on Philox4x32-10) and
device sorts, with ONE host read per hop (the block's edge count).  The block leaves as CSR, so the
aggregate consumes it through ``Engine.plan_from_rowptr`` with no sort and no further sync.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import engine as _engine
from .ops import _ptr


_BIG = 1 << 62


def sample_adj(rowptr, col, idx, num_neighbors, replace=False, eng=None, first_pos=None):
    """Sample ``num_neighbors`` in-neighbours of every row in ``idx`` (all of them if negative).

    ``first_pos``: optional int64 scratch of one entry per graph node, filled with ``_BIG`` (NeighborSampler
    keeps one): the relabelling then runs on it without sorting (first occurrence of every node by a
    scatter-min, ids by a prefix sum over the first occurrences) and hands it back reset."""
    eng = eng or _engine()
    dev = eng._dev(rowptr, col, idx)
    rowptr = rowptr.contiguous().to(torch.int64)
    col = col.contiguous().to(torch.int64)
    idx = idx.contiguous().to(torch.int64).reshape(-1)
    B = int(idx.shape[0])
    st = eng._stream(dev)
    deg = torch.empty(B, dtype=torch.int64, device=dev)
    eng._check(eng.lib.ggl_sample_count(_ptr(rowptr), _ptr(idx), B, int(num_neighbors), int(bool(replace)),
                                        _ptr(deg), st))
    out_rowptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=out_rowptr[1:])
    E = int(out_rowptr[-1]) if B > 0 else 0  # the one host read of this hop
    e_pos = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    nbr = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    eng._check(eng.lib.ggl_sample_pick(_ptr(rowptr), _ptr(col), _ptr(idx), B, int(num_neighbors),
                                       int(bool(replace)), _ptr(out_rowptr), _ptr(eng._rng_state(dev)),
                                       _ptr(e_pos), _ptr(nbr), st))
    e_pos, nbr = e_pos[:E], nbr[:E]
    # relabel: seeds keep 0..B-1 (sample.cpp:24-29), new nodes follow in first-seen order (:48-51)
    cat = torch.cat([idx, nbr])
    if first_pos is not None:
        ar = torch.arange(cat.shape[0], device=dev)
        first_pos.scatter_reduce_(0, cat, ar, "amin", include_self=True)   # first position of every node met
        fp = first_pos[cat]
        is_first = fp == ar
        new_id = torch.cumsum(is_first, 0) - 1                              # ids in first-seen order
        out_n_id = cat[is_first]
        local = new_id[fp[B:]]
        first_pos[cat] = _BIG                                               # hand the scratch back clean
    else:
        uniq, inv = torch.unique(cat, return_inverse=True)
        first = torch.full((uniq.shape[0],), cat.shape[0], dtype=torch.int64, device=dev)
        first.scatter_reduce_(0, inv, torch.arange(cat.shape[0], device=dev), "amin", include_self=True)
        order = torch.argsort(first)                      # unique ids by first appearance
        rank = torch.empty_like(order)
        rank[order] = torch.arange(order.shape[0], device=dev)
        out_n_id = uniq[order]
        local = rank[inv[B:]]
    # every row's columns ascending by local id (sample.cpp:112-118)
    if E > 0:
        row = torch.repeat_interleave(torch.arange(B, device=dev), deg, output_size=E)  # size known: no sync
        perm = torch.empty(E, dtype=torch.int32, device=dev)
        wsb = eng.lib.ggl_sort_edges_workspace_bytes(E, int(out_n_id.shape[0]))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        eng._check(eng.lib.ggl_sort_edges(_ptr(row), _ptr(local.contiguous()), E, int(out_n_id.shape[0]),
                                          _ptr(perm), _ptr(ws), wsb, st))
        p = perm.long()
        local, e_pos = local[p], e_pos[p]
    return out_rowptr, local, out_n_id, e_pos


@dataclass
class EdgeIndex:  # loader/neighbor_sampler.py:12-16
    edge_index: torch.Tensor
    e_id: Optional[torch.Tensor]
    size: Tuple[int, int]
    rowptr: Optional[torch.Tensor] = None   # extra: the block's CSR row pointer (dst-major), for plan_from_rowptr
    fanout: Optional[int] = None


class NeighborSampler:
    """``loader/neighbor_sampler.py:29-112`` without the DataLoader plumbing: build the transposed CSR
    once, then ``sample(batch)`` returns ``(batch, n_id, adjs)`` with ``adjs`` outermost hop first."""

    def __init__(self, edge_index, sample_lists, num_nodes=None, eng=None):
        self.eng = eng or _engine()
        self.sizes = list(sample_lists)
        ei = edge_index.contiguous().to(torch.int64)
        if num_nodes is None:
            num_nodes = int(ei.max()) + 1
        self.num_nodes = int(num_nodes)
        # adj_t = SparseGraph(row=src, col=dst, value=arange(E)).t(): rows = destination, cols = source
        plan = self.eng.seg_plan(ei[1].contiguous(), self.num_nodes)
        self.rowptr = plan.rowptr
        self.col = ei[0] if plan.perm is None else ei[0][plan.perm.long()]
        self.value = (torch.arange(ei.shape[1], device=ei.device) if plan.perm is None else plan.perm.long())
        self._first_pos = torch.full((self.num_nodes,), _BIG, dtype=torch.int64, device=ei.device)  # relabel scratch

    def sample(self, batch):
        batch = torch.as_tensor(batch, device=self.rowptr.device, dtype=torch.int64).reshape(-1)
        n_id, adjs = batch, []
        for size in self.sizes:
            n_dst = int(n_id.shape[0])
            rowptr, col, n_id, e_pos = sample_adj(self.rowptr, self.col, n_id, size, replace=False, eng=self.eng,
                                                  first_pos=self._first_pos)
            row = torch.repeat_interleave(torch.arange(n_dst, device=col.device), rowptr[1:] - rowptr[:-1],
                                          output_size=int(col.shape[0]))
            block = torch.stack([col, row])
            adjs.append(EdgeIndex(block, self.value[e_pos], (int(n_id.shape[0]), n_dst), rowptr, size))
            if size >= 0:  # rows hold <= fan-out entries: the aggregate's plan needs neither sort nor sync
                plan = self.eng.plan_from_rowptr(rowptr, int(col.shape[0]), max_len=size)
                self.eng.adopt_plan(block[1], n_dst, plan)
        adjs = adjs[0] if len(adjs) == 1 else adjs[::-1]
        return batch, n_id, adjs


__all__ = ["sample_adj", "NeighborSampler", "EdgeIndex"]
