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
EXACT_HOP_MAX_SLOTS = 1 << 26   # seeds x fan-out up to which NeighborSampler.sample runs a hop at worst-case capacity (4 GB)


def sample_adj(rowptr, col, idx, num_neighbors, replace=False, eng=None, first_pos=None):
    """Sample ``num_neighbors`` in-neighbours of every row in ``idx`` (all of them if negative).

    ``first_pos``: optional int64 scratch of one entry per graph node, filled with ``_BIG`` (NeighborSampler
    keeps one): the relabelling then runs on it without sorting (first occurrence of every node by a
    scatter-min, ids by a prefix sum over the first occurrences) and hands it back reset."""
    eng = eng or _engine(rowptr)   # (CPU tensors: the host build, as the reference's c_sample_adj serves them)
    dev = eng._dev(rowptr, col, idx)
    rowptr = rowptr.contiguous().to(torch.int64)
    col = col.contiguous().to(torch.int64)
    idx = idx.contiguous().to(torch.int64).reshape(-1)
    B = int(idx.shape[0])
    st = eng._stream(dev)
    deg = torch.empty(B, dtype=torch.int64, device=dev)
    # (a seed outside [0, N) gets no neighbours inside the kernels — no out-of-bounds read — and stays in n_id,
    #  where the caller's feature gather x[n_id] reports it)
    eng._check(eng.lib.ggl_sample_count(_ptr(rowptr), _ptr(idx), B, int(rowptr.shape[0]) - 1, int(num_neighbors),
                                        int(bool(replace)), _ptr(deg), st))
    out_rowptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=out_rowptr[1:])
    E = int(out_rowptr[-1]) if B > 0 else 0  # the one host read of this hop
    e_pos = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    nbr = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    eng._check(eng.lib.ggl_sample_pick(_ptr(rowptr), _ptr(col), _ptr(idx), B, int(num_neighbors),
                                       int(bool(replace)), _ptr(out_rowptr), _ptr(eng._rng_state(dev)),
                                       _ptr(e_pos), _ptr(nbr), st))
    e_pos, nbr = e_pos[:E], nbr[:E]
    # relabel: the seeds keep 0..B-1 VERBATIM — a seed listed twice stays twice in n_id, and a neighbour equal to
    # it maps to its LAST position (operator[] overwrite, sample.cpp:24-29) — new nodes follow in first-seen
    # order (:48-51).  Keys: seed i -> -(i + 1), sampled neighbour q -> B + q; the minimum key per node is the
    # last seed position if the node is a seed, else its first sampled occurrence.
    cat = torch.cat([idx, nbr])
    key = torch.cat([torch.arange(-1, -B - 1, -1, device=dev), torch.arange(B, B + E, device=dev)])
    if first_pos is not None:
        first_pos.scatter_reduce_(0, cat, key, "amin", include_self=True)
        fp = first_pos[cat]
        first_pos[cat] = _BIG                                               # hand the scratch back clean
    else:
        uniq, inv = torch.unique(cat, return_inverse=True)
        first = torch.full((uniq.shape[0],), _BIG, dtype=torch.int64, device=dev)
        first.scatter_reduce_(0, inv, key, "amin", include_self=True)
        fp = first[inv]
    is_first = fp == key
    is_first[:B] = True
    new_id = torch.cumsum(is_first, 0) - 1                                  # ids in first-seen order
    out_n_id = cat[is_first]
    # key -> local id in one gather: keys >= 0 index new_id, a seed key -(i + 1) wraps to the tail, which holds i
    local = torch.cat([new_id, torch.arange(B - 1, -1, -1, device=dev)])[fp[B:]]
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


def neighbor_sample(colptr, row, input_nodes, fanouts, replace=False, eng=None):
    """The multi-hop sampler behind ``cuda_torch_neighbor_sample`` (ops/sparse/cuda/neighbor_sample.cu:744-778 ->
    cu_neighbor_sample :634-741), chained on the device from the per-hop kernels:

    * hop l samples ``min(degree, fanouts[l])`` distinct in-edges (all of them for -1) of every node ADDED by hop
      l - 1 (hop 0: ``input_nodes``) — cu_neighbor_sample_one_hop :198-306, frontier = ``input_nodes +
      last_num_input_node`` :647-650; the edge ids are positions in ``row`` (``row_offset + colptr[node]`` :131),
      frontier node by frontier node;
    * the nodes those edges lead to (``row[eid]`` :308-318) that are not in the node list yet are appended to it in
      ASCENDING id order (get_new_input_nodes :436-532: a sort of old / new ids tagged in the low bit, new = odd, first
      of its value and not preceded by its own even twin); no new node ends the loop early (:665-668);
    * returns ``[sample_cols, sample_rows, sample_nodes, sample_edges]`` (:729-733): the node list, the edge ids of all
      hops concatenated (:706-718), and per edge the position in the node list of the node it was sampled FOR (the owner
      of that stretch of ``row``, kernal_get_col :362-378) and of the node it leads to (kernal_get_row :380-396).

    ``replace=True`` leaves the reference's hop undefined (:217-218 is an empty branch); here it draws with replacement.
    The draws come from the engine's device RNG (Floyd's algorithm on Philox4x32-10), not from ``random_seed`` + cuRAND's
    reservoir-with-atomicMax (:103-135) — same contract (distinct in-edges, uniformly), other bits.  One host read per
    hop (the hop's edge count)."""
    eng = eng or _engine(colptr)
    dev = eng._dev(colptr, row, input_nodes)
    colptr = colptr.contiguous().to(torch.int64)
    row = row.contiguous().to(torch.int64)
    nodes = input_nodes.contiguous().to(torch.int64).reshape(-1)
    fan = [int(f) for f in (fanouts.reshape(-1).tolist() if isinstance(fanouts, torch.Tensor) else list(fanouts))]
    n_graph = int(colptr.shape[0]) - 1
    st = eng._stream(dev)
    frontier, f_off = nodes, 0                      # the nodes the next hop samples for, and where they sit in `nodes`
    eids, owners = [], []
    for fanout in fan:
        B = int(frontier.shape[0])
        if B == 0:
            break
        deg = torch.empty(B, dtype=torch.int64, device=dev)
        eng._check(eng.lib.ggl_sample_count(_ptr(colptr), _ptr(frontier), B, n_graph, fanout, int(bool(replace)), _ptr(deg), st))
        ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(deg, 0, out=ptr[1:])
        E = int(ptr[-1])                             # the one host read of this hop
        e_pos = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
        nbr = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
        eng._check(eng.lib.ggl_sample_pick(_ptr(colptr), _ptr(row), _ptr(frontier), B, fanout, int(bool(replace)), _ptr(ptr),
                                           _ptr(eng._rng_state(dev)), _ptr(e_pos), _ptr(nbr), st))
        e_pos, nbr = e_pos[:E], nbr[:E]
        eids.append(e_pos)
        owners.append(torch.repeat_interleave(torch.arange(f_off, f_off + B, device=dev), deg, output_size=E))
        new = torch.unique(nbr)                      # ascending
        new = new[~torch.isin(new, nodes)]
        if new.numel() == 0:
            break
        f_off = int(nodes.shape[0])
        nodes = torch.cat([nodes, new])
        frontier = new
    edges = torch.cat(eids) if eids else torch.empty(0, dtype=torch.int64, device=dev)
    cols = torch.cat(owners) if owners else torch.empty(0, dtype=torch.int64, device=dev)
    # position of every sampled edge's far end in the node list (each is there by construction)
    order = torch.argsort(nodes, stable=True)
    srt = nodes[order]
    far = row[edges]
    rows = order[torch.searchsorted(srt, far).clamp(max=max(int(nodes.shape[0]) - 1, 0))] if edges.numel() else edges.clone()
    return [cols, rows, nodes, edges]


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
        self.eng = eng or _engine(edge_index)   # (CPU tensors: the host build, like the reference's CPU sampler)
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

    def _hop_exact(self, seeds, fanout):
        """One sampled hop with exact output shapes: the static-shape kernels (ggl_sample_hop: ~15 launches, relabelling
        included) at worst-case capacity, then ONE host read of the counts to cut the buffers to size — instead of the
        count / pick kernels plus ~80 torch launches of relabelling and sorting of `sample_adj`."""
        eng, dev = self.eng, self.rowptr.device
        b = int(seeds.shape[0])
        e_cap, s_cap = max(b * fanout, 1), b + max(b * fanout, 1)
        rowptr = torch.empty(b + 1, dtype=torch.int64, device=dev)
        col = torch.empty(e_cap, dtype=torch.int32, device=dev)
        e_pos = torch.empty(e_cap, dtype=torch.int64, device=dev)
        nid = torch.empty(s_cap, dtype=torch.int64, device=dev)
        counts = torch.zeros(3, dtype=torch.int64, device=dev)
        n_seeds = torch.full((1,), b, dtype=torch.int64, device=dev)
        wsb = eng.lib.ggl_sample_hop_workspace_bytes(b, e_cap)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        eng._check(eng.lib.ggl_sample_hop(_ptr(self.rowptr), _ptr(self.col), _ptr(seeds), _ptr(n_seeds), b,
                                          self.num_nodes, int(fanout), e_cap, s_cap, _ptr(eng._rng_state(dev)),
                                          _ptr(self._first_pos), _ptr(rowptr), _ptr(col), _ptr(e_pos), _ptr(nid),
                                          _ptr(counts), _ptr(ws), wsb, eng._stream(dev)))
        nn, ne, _ = counts.tolist()   # the one host read of this hop
        return rowptr, col[:ne].long(), nid[:nn], e_pos[:ne]

    def sample(self, batch):
        batch = torch.as_tensor(batch, device=self.rowptr.device, dtype=torch.int64).reshape(-1)
        n_id, adjs = batch, []
        for size in self.sizes:
            n_dst = int(n_id.shape[0])
            # the static-shape hop allocates WORST-CASE buffers (~60 bytes per seed x fan-out slot) before the counts are
            # known: fine for mini-batches, not for a layer-wise inference batch of tens of millions of seeds, which
            # keeps the count-then-pick path whose buffers are sized by the actual min(deg, fan-out)
            if size > 0 and n_dst > 0 and n_dst * size <= EXACT_HOP_MAX_SLOTS:
                rowptr, col, n_id, e_pos = self._hop_exact(n_id.contiguous(), size)
            else:
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


class Block:
    """One sampled hop with FIXED capacities and device-side sizes (ggl_sample_hop): a CSR over `n_dst_cap`
    destination rows (the hop's seeds, of which counts-of-the-previous-hop are valid; the rest are empty) with
    LOCAL int32 source ids < n_src_cap, `e_cap` = n_dst_cap * fanout edge slots of which rowptr[n_dst_cap] are
    used.  `counts` (device int64 [2]) = {nodes in n_id, sampled edges}.  Nothing about a Block needs a host
    read, so the aggregate over it and its backward capture into a hipGraph."""

    def __init__(self, eng, rowptr, col, e_pos, counts, n_dst_cap, n_src_cap, fanout):
        from .ops import SegPlan

        self.eng, self.rowptr, self.col, self.e_pos, self.counts = eng, rowptr, col, e_pos, counts
        self.n_dst_cap, self.n_src_cap, self.fanout = int(n_dst_cap), int(n_src_cap), int(fanout)
        self.e_cap = int(col.shape[0])
        self.size = (self.n_src_cap, self.n_dst_cap)           # EdgeIndex.size convention: (sources, targets)
        p = SegPlan()
        p.N, p.E, p.rowptr, p.perm, p.is_sorted = self.n_dst_cap, self.e_cap, rowptr, None, True
        p.max_len, p.chunk, p.n_long, p.n_chunks = self.fanout, 1 << 62, 0, 0
        p.long_rows = p.chunk_ptr = p.row_order = None
        p.device, p.uid = rowptr.device, -1
        self.plan = p
        self._T = None

    def transposed(self):
        """(planT, dstT): the block's CSC — for every source row the destination rows of its edges — built on the
        device without a host read (ggl_block_transpose); hub sources are walked in one piece (no long-row table:
        its size would have to be read back)."""
        if self._T is None:
            from .ops import SegPlan

            eng, dev = self.eng, self.rowptr.device
            rowptrT = torch.empty(self.n_src_cap + 1, dtype=torch.int64, device=dev)
            dstT = torch.empty(max(self.e_cap, 1), dtype=torch.int32, device=dev)
            wsb = eng.lib.ggl_block_transpose_workspace_bytes(self.e_cap, self.n_src_cap)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            eng._check(eng.lib.ggl_block_transpose(_ptr(self.rowptr), _ptr(self.col), self.n_dst_cap, self.n_src_cap,
                                                   self.e_cap, _ptr(rowptrT), _ptr(dstT), _ptr(ws), wsb,
                                                   eng._stream(dev)))
            p = SegPlan()
            p.N, p.E, p.rowptr, p.perm, p.is_sorted = self.n_src_cap, self.e_cap, rowptrT, None, True
            p.max_len, p.chunk, p.n_long, p.n_chunks = self.e_cap, 1 << 62, 0, 0
            p.long_rows = p.chunk_ptr = p.row_order = None
            p.device, p.uid = dev, -1
            self._T = (p, dstT)
        return self._T


class BlockSampler:
    """NeighborSampler with static shapes: `sample(seeds)` runs every hop on the device with fixed-capacity
    buffers and device-side counts — no host read, ~10 launches per hop — so a training step built on it
    captures into one hipGraph (trainer.SAGEBlockTrainer).  Same blocks as loader/neighbor_sampler.py:74-109 +
    sample.cpp:10-135 produce (seeds first and verbatim, new nodes in first-seen order, rows sorted by local id,
    min(deg, fanout) distinct neighbours per row), padded to capacity."""

    def __init__(self, edge_index, sample_lists, num_nodes=None, eng=None):
        self.eng = eng or _engine(edge_index)   # (CPU tensors: the host build, like the reference's CPU sampler)
        self.sizes = [int(s) for s in sample_lists]
        if any(s <= 0 for s in self.sizes):
            raise ValueError("BlockSampler needs positive fan-outs (use NeighborSampler for full neighbourhoods)")
        ei = edge_index.contiguous().to(torch.int64)
        if num_nodes is None:
            num_nodes = int(ei.max()) + 1
        self.num_nodes = int(num_nodes)
        plan = self.eng.seg_plan(ei[1].contiguous(), self.num_nodes)
        self.rowptr = plan.rowptr
        self.col = ei[0].contiguous() if plan.perm is None else ei[0][plan.perm.long()].contiguous()
        self.value = (torch.arange(ei.shape[1], device=ei.device) if plan.perm is None else plan.perm.long())
        self._first_pos = torch.full((self.num_nodes,), _BIG, dtype=torch.int64, device=ei.device)
        self._overflow = torch.zeros((), dtype=torch.int64, device=ei.device)
        self._n_full = {}

    def capacities(self, batch_size):
        """Worst-case [(n_src_cap, e_cap)] per hop, innermost (the seeds' own block) first."""
        caps, b = [], int(batch_size)
        for f in self.sizes:
            caps.append((b + b * f, b * f))
            b = b + b * f
        return caps

    def calibrate(self, batch_size, trials=8, slack=1.25, seed=0):
        """Capacities from measured batches: `trials` random batches at worst-case capacity, then
        slack x the largest counts seen, rounded up to 256 (the worst case — every seed with `fanout`
        distinct, previously unseen neighbours — is several times what a power-law graph produces, and every
        dense layer would run on the padding).  A batch that still exceeds them raises the block's overflow
        flag (`overflow_count()`); the caller re-runs it with `capacities(batch_size)`."""
        dev = self.rowptr.device
        g = torch.Generator(device=dev).manual_seed(seed)
        mx = [[0, 0] for _ in self.sizes]
        for _ in range(trials):
            sd = torch.randint(0, self.num_nodes, (batch_size,), generator=g, device=dev)
            _, blocks, _ = self.sample(sd)
            for h, blk in enumerate(blocks[::-1]):
                c = blk.counts.tolist()
                mx[h] = [max(mx[h][0], c[0]), max(mx[h][1], c[1])]
        caps, b = [], int(batch_size)
        for (f, (nn, ne)) in zip(self.sizes, mx):
            r256 = lambda v: -(-int(v) // 256) * 256 if v >= 4096 else -(-int(v) // 16) * 16  # noqa: E731
            e_cap = max(1, min(b * f, r256(ne * slack)))
            s_cap = max(b, min(b + e_cap, r256(nn * slack)))
            caps.append((s_cap, e_cap))
            b = s_cap
        return caps

    def overflow_count(self):
        """How many sampled hops hit a capacity since the sampler was built (one host read)."""
        return int(self._overflow)

    def sample(self, seeds, n_seeds=None, caps=None):
        """seeds: int64 device tensor [B]; n_seeds: optional device int64 [1] (<= B valid seeds); caps: optional
        [(n_src_cap, e_cap)] per hop, innermost first (default: the worst case).
        Returns (n_id [cap], blocks outermost hop first, counts of the outermost hop)."""
        eng = self.eng
        dev = self.rowptr.device
        seeds = seeds.to(device=dev, dtype=torch.int64).contiguous().reshape(-1)
        if n_seeds is None:   # "every seed is valid": one constant per batch size, made once (a fill per call otherwise)
            key = (int(seeds.shape[0]), str(dev))
            n_seeds = self._n_full.get(key)
            if n_seeds is None:
                n_seeds = self._n_full[key] = torch.full((1,), seeds.shape[0], dtype=torch.int64, device=dev)
        st = eng._stream(dev)
        blocks = []
        cur, n_cur = seeds, n_seeds
        for h, f in enumerate(self.sizes):
            b_cap = int(cur.shape[0])
            s_cap, e_cap = caps[h] if caps is not None else (b_cap + b_cap * f, b_cap * f)
            s_cap, e_cap = int(s_cap), int(e_cap)
            rowptr = torch.empty(b_cap + 1, dtype=torch.int64, device=dev)
            col = torch.empty(e_cap, dtype=torch.int32, device=dev)
            e_pos = torch.empty(e_cap, dtype=torch.int64, device=dev)
            nid = torch.empty(s_cap, dtype=torch.int64, device=dev)
            counts = torch.empty(3, dtype=torch.int64, device=dev)   # (all three are assigned by the hop's kernels)
            wsb = eng.lib.ggl_sample_hop_workspace_bytes(b_cap, e_cap)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            # (the running overflow count is kept by the hop's own kernels: `self._overflow += counts[2]` was a launch per hop)
            eng._check(eng.lib.ggl_sample_hop_ex(_ptr(self.rowptr), _ptr(self.col), _ptr(cur), _ptr(n_cur), b_cap,
                                                 self.num_nodes, f, e_cap, s_cap, _ptr(eng._rng_state(dev)), _ptr(self._first_pos),
                                                 _ptr(rowptr), _ptr(col), _ptr(e_pos), _ptr(nid), _ptr(counts), _ptr(ws),
                                                 wsb, _ptr(self._overflow), st))
            blk = Block(eng, rowptr, col, e_pos, counts, b_cap, s_cap, f)
            blk.n_id, blk.seeds, blk.n_seeds = nid, cur, n_cur   # global ids of its rows / of its seeds
            blocks.append(blk)
            cur, n_cur = nid, counts[0:1]
        return cur, blocks[::-1], blocks[-1].counts


__all__ = ["sample_adj", "NeighborSampler", "EdgeIndex", "Block", "BlockSampler"]
