"""Locality-aware node ordering for the 1-D partition ("locality-aware partition ... mandatory"):
the halo a rank receives per layer is exactly the edge cut of its node range, so nodes that are connected
should sit in the same range.  The reference has no counterpart (no distributed code at all).

`cluster_order` finds `clusters` (>= parts) balanced communities by size-constrained label propagation and
returns a relabelling that makes every community a contiguous id range; `dist.balanced_bounds` then cuts the
relabelled graph as usual.  Each sweep is ONE SpMM of this library on the one-hot label matrix
(scores[i, c] = edges from i into community c) plus an argmax — the aggregate it serves is also its engine.

What it can and cannot do is a property of the graph: on a graph with community structure (planted classes,
real co-purchase / citation graphs) the halo shrinks by the fraction of intra-community edges; on R-MAT, which
has none, nothing does (tools/halo_stats.py measures both)."""
import torch

from . import engine as _engine


def arrange_communities(q, device=None):
    """A linear arrangement of the communities of a quotient graph (`q` [C, C]: edges between communities) that puts
    strongly connected communities next to each other, by RECURSIVE SPECTRAL BISECTION: order the communities along
    the Fiedler vector of the normalised Laplacian, cut that order where the normalised cut is smallest, recurse into
    both sides (C is at most ~1000: dense eigendecompositions on the host, O(C^3) in total).  A hierarchy (classes
    inside super-classes inside ...) comes out as nested contiguous ranges, so a contiguous cut of the final order into
    P parts severs the weakest links it can.  Returns `pos` int64 [C]: position of every community (empty
    communities last).  Deterministic (eigenvector signs are fixed), so every rank computes the same arrangement."""
    q = q.detach().double().to(device if device is not None else "cpu")   # (thousands of communities: the eigen-
    C = q.shape[0]                                                          #  decompositions run on the GPU)
    q = q + q.t()
    q.fill_diagonal_(0.0)
    live = torch.nonzero(q.sum(1) > 0).reshape(-1).tolist()
    live_set = set(live)
    dead = [c for c in range(C) if c not in live_set]

    def split(nodes):
        """One bisection step: ('leaf', nodes) when the group is final, else ('split', left, right) — both sides still to
        be ordered, left before right.  ('tail', keep, lone): `lone` (no link into the group) goes after `keep`."""
        if len(nodes) <= 2:
            return ("leaf", nodes)
        idx = torch.tensor(nodes, device=q.device)
        sub = q[idx][:, idx]
        d = sub.sum(1)
        lone = [nodes[i] for i in torch.nonzero(d <= 0).reshape(-1).tolist()]
        if lone:   # communities with no link into this group: to the end, the rest is ordered on its own
            lone_set = set(lone)
            keep = [n for n in nodes if n not in lone_set]
            return ("tail", keep, lone) if keep else ("leaf", nodes)
        dis = d.pow(-0.5)
        lap = torch.eye(len(nodes), dtype=torch.float64, device=q.device) - dis.unsqueeze(1) * sub * dis.unsqueeze(0)
        _, vec = torch.linalg.eigh(lap)
        f = vec[:, 1] * dis
        if float(f[torch.argmax(f.abs())]) < 0:
            f = -f
        o = torch.argsort(f, stable=True)
        so = sub[o][:, o]
        vol = so.sum(1)
        # sweep: cut after position k; cut weight = links from the first k + 1 to the rest
        csum_vol = torch.cumsum(vol, 0)
        inside = torch.cumsum(torch.cumsum(so, 1).diagonal() * 2 - 0, 0)     # 2 x links among the first k + 1 (diag is 0)
        cut = csum_vol - inside
        denom = torch.minimum(csum_vol, csum_vol[-1] - csum_vol).clamp(min=1e-12)
        score = (cut / denom)[:-1]
        k = int(torch.argmin(score)) + 1
        return ("split", [nodes[i] for i in o[:k].tolist()], [nodes[i] for i in o[k:].tolist()])

    def order(nodes):
        # an explicit stack instead of recursion: a run of unbalanced cuts (k = 1 on star / chain quotient graphs of up to
        # 8192 communities) is as deep as it is long, far past Python's frame limit
        out, stack = [], [("todo", nodes)]
        while stack:
            tag, grp = stack.pop()
            if tag == "done":
                out.extend(grp)
                continue
            r = split(grp)
            if r[0] == "leaf":
                out.extend(r[1])
            elif r[0] == "tail":
                stack.append(("done", r[2]))
                stack.append(("todo", r[1]))
            else:
                stack.append(("todo", r[2]))
                stack.append(("todo", r[1]))
        return out

    seq = order(live) + dead
    pos = torch.empty(C, dtype=torch.int64)
    pos[torch.tensor(seq, dtype=torch.int64)] = torch.arange(C)
    return pos


def _sweep_by_sorting(lab_src, dst, lab, C, cap, size):
    """One propagation sweep without the [N, C] one-hot matrix: the (node, neighbour's community) pairs of all edges are
    sorted and run-length counted, so the cost is one sort of E keys whatever the number of communities (the SpMM form
    moves N x C floats per sweep: 20 GB at 2 000 labels on the products-sized graph).  `lab_src` [E] = community of every
    edge's source, `dst` [E] = its destination among the `lab.numel()` nodes whose labels `lab` holds (all nodes, or a
    rank's own), `size` [C] the GLOBAL community sizes.  Returns the label every node would adopt (same scoring as the
    dense form: neighbour count, a small pull towards smaller communities, full communities closed to newcomers, ties
    stay put)."""
    dev = lab.device
    N = int(lab.shape[0])
    ar = torch.arange(N, device=dev)
    key = torch.cat([dst * C + lab_src, ar * C + lab])            # + every node's own label (count 0 if no neighbour has it)
    uk, cnt = torch.unique(key, return_counts=True)
    node, l = uk // C, uk % C
    own = l == lab[node]
    sc = (cnt - own.to(cnt.dtype)).float() - 1e-3 * (size[l] / cap)   # (the appended own-label entry is not a neighbour)
    full = size[l] >= cap
    sc = torch.where(full & ~own, torch.full_like(sc, -1.0), sc)
    sc = sc + own.float() * (0.5 + full.float() * 2.0)
    mx = torch.full((N,), -2.0, device=dev).scatter_reduce_(0, node, sc, "amax", include_self=True)
    cand = torch.where(sc >= mx[node], l, torch.full_like(l, C))
    # ties: the current label if it is among the best, else the smallest label id (deterministic)
    best = torch.full((N,), C, dtype=torch.int64, device=dev).scatter_reduce_(0, node, cand, "amin", include_self=True)
    keeps = torch.zeros(N, dtype=torch.bool, device=dev)
    keeps[node[own & (sc >= mx[node])]] = True
    return torch.where(keeps, lab, best)


def _admit(new, lab, C, room, before=None):
    """Movers are admitted to their target community only up to its free `room`, lowest node id first (deterministic).
    `before` [C] (distributed runs): movers to each community on lower ranks — ranks own ascending id ranges, so
    "lowest id first" over the whole graph is "lower ranks first, then local id"."""
    N, dev = int(lab.shape[0]), lab.device
    ar = torch.arange(N, device=dev)
    move = new != lab
    tgt = torch.where(move, new, torch.full_like(new, C))
    order = torch.argsort(tgt * N + ar)                             # movers grouped by target community
    st = tgt[order]
    first = torch.searchsorted(st, torch.arange(C + 1, device=dev))
    pos = torch.arange(N, device=dev) - first[st.clamp(max=C)]
    stc = st.clamp(max=C - 1)
    if before is not None:
        pos = pos + before[stc]
    ok = (st < C) & (pos < room[stc])
    out = lab.clone()
    out[order[ok]] = st[ok]
    return out, move


def cluster_order(edge_index, num_nodes, clusters=64, sweeps=20, seed=0, eng=None, balance=4.0, arrange=True, update=1.0,
                  method="auto"):
    """Returns (`rank`, `label`): `rank` int64 [N] = new id of every node (communities contiguous, ids stable
    inside a community), `label` the community of every node.  Size-capped label propagation: a node adopts the
    community most of its neighbours are in unless that community already holds `balance` x the average
    (synchronous sweeps, ties keep the current label).  Use several times more `clusters` than partitions: the
    cap only has to stop a collapse into one giant community — `balanced_bounds` cuts the community-sorted order
    by edge count, and a cut through the middle of a community costs just that community's split (measured on a
    planted-partition graph: 32 labels for 8 planted classes recover them, local-source share 0.13 -> 0.78 at
    P = 8; with exactly 8 tightly balanced labels the propagation stalls at 0.28).  `arrange`: the communities are
    laid out by recursive spectral bisection of their quotient graph (`arrange_communities`) instead of by label id.
    `method`: "spmm" (each sweep one SpMM of this library on the one-hot label matrix), "sort" (`_sweep_by_sorting`:
    no N x C matrix, for thousands of communities), "auto" = sort above 1024 communities."""
    eng = eng or _engine()
    dev = edge_index.device
    N, C = int(num_nodes), int(clusters)
    g = torch.Generator(device=dev).manual_seed(seed)
    lab = torch.randint(0, C, (N,), generator=g, device=dev)
    by_sort = method == "sort" or (method == "auto" and C > 1024)
    gp = None if by_sort else eng.graph_plan(edge_index, N)
    src, dst = edge_index[0], edge_index[1]
    cap = balance * N / C
    ar = torch.arange(N, device=dev)
    for _ in range(sweeps):
        size = torch.bincount(lab, minlength=C).float()
        if by_sort:
            new = _sweep_by_sorting(lab[src], dst, lab, C, cap, size)
        else:
            onehot = torch.zeros((N, C), dtype=torch.float32, device=dev)
            onehot[ar, lab] = 1.0
            with torch.no_grad():
                score = eng.spmm(gp, None, onehot)                     # [N, C]: neighbours per community
            score = score - 1e-3 * (size / cap).unsqueeze(0)            # tie-break towards the smaller community
            score[:, size >= cap] = -1.0                                # full communities accept nobody new ...
            score[ar, lab] += 0.5 + (size[lab] >= cap).float() * 2.0    # ... but keep their members; ties stay put
            new = score.argmax(1)
        if update < 1.0:   # damped synchronous sweeps: a random share of the nodes may move (fewer two-cycles)
            new = torch.where(torch.rand(N, generator=g, device=dev) < update, new, lab)
        # admit movers only up to each community's free room (lowest node id first: deterministic)
        lab, _ = _admit(new, lab, C, (cap - size).clamp(min=0))
    pos = torch.arange(C, device=dev)
    if arrange and C > 2:
        # communities in the order of the quotient graph's Fiedler vector, so that neighbouring id ranges hold
        # communities that exchange many edges (label ids themselves carry no meaning)
        q = torch.bincount(lab[dst] * C + lab[src], minlength=C * C).view(C, C).double()    # edges between communities
        pos = arrange_communities(q, device=dev if (C > 1500 and dev.type == "cuda") else None).to(dev)
    rank = torch.empty(N, dtype=torch.int64, device=dev)
    rank[torch.argsort(pos[lab] * N + ar)] = ar
    return rank, lab


class HaloIndex:
    """Which entries of a per-node vector a rank's edges read from other ranks, and the one all-to-all that fetches
    them: the id-only counterpart of `dist.PartitionedGraph`'s halo bookkeeping, for passes that run BEFORE the
    partition is final (label sweeps, renaming).  `src` global ids, the rank owns [lo, hi)."""

    def __init__(self, src, lo, hi, bounds, comm):
        dev = src.device
        self.comm, self.n_local = comm, hi - lo
        rem = (src < lo) | (src >= hi)
        self.halo_ids = torch.unique(src[rem])
        bt = torch.tensor(bounds[1:-1], device=dev, dtype=torch.int64)
        owner = torch.searchsorted(bt, self.halo_ids, right=True)          # ascending ids -> already grouped by owner
        self.recv_counts = torch.bincount(owner, minlength=comm.world).tolist()
        wanted = comm.route(self.halo_ids, owner)                            # ids the others read from me, by rank
        sc = torch.tensor(self.recv_counts, dtype=torch.int64, device=dev)
        self.send_counts = comm.all_to_all_counts(sc).tolist()
        self.send_idx = wanted - lo
        self.src_idx = torch.where(rem, self.n_local + torch.searchsorted(self.halo_ids, src), src - lo)

    def gather(self, v):
        """[v ; v at the halo ids] for a vector `v` over the rank's own nodes."""
        if self.comm.world == 1 and not self.comm.always:
            return v
        import torch.distributed as dist

        out = torch.empty(sum(self.recv_counts), dtype=v.dtype, device=v.device)
        dist.all_to_all_single(out, v[self.send_idx].contiguous(), self.recv_counts, self.send_counts,
                               group=self.comm.group)
        return torch.cat([v, out])


def cluster_order_distributed(g, comm, clusters=64, sweeps=20, seed=0, balance=4.0, arrange=True):
    """`cluster_order` for a graph no single rank holds: `g` is the rank's share as `synth.rmat_partitioned` returns it
    (edges into the nodes it owns).  Per sweep the ranks exchange the labels of their halo sources (one all-to-all of
    int64 ids — 8 bytes per halo node against the 400 of a feature row), run `_sweep_by_sorting` on their own edges,
    and agree on the community sizes and on the order movers are admitted in (an all-gather of [C] counters).
    Deterministic and IDENTICAL to the single-process result for the same seed and method="sort": same initial
    labels (one seeded generator, every rank keeps its slice), same scores, and the admission order "lowest node id
    first" is "lower ranks first" because ranks own ascending ranges.  Returns (`new_id` [n_local]: new global id of
    every own node, `label` [n_local])."""
    src, dstl, bounds, N = g["src"], g["dst"], g["bounds"], int(g["num_nodes"])
    dev = src.device
    C = int(clusters)
    lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
    gen = torch.Generator(device=dev).manual_seed(seed)
    lab = torch.randint(0, C, (N,), generator=gen, device=dev)[lo:hi].clone()
    real = src != dstl + lo                                   # (the shares carry the self-loops; the sweeps ignore them,
    src, dstl = src[real], dstl[real]                         #  as `synth.full_graph_partitioned` does)
    hx = HaloIndex(src, lo, hi, bounds, comm)
    cap = balance * N / C
    for _ in range(sweeps):
        size = comm.all_reduce(torch.bincount(lab, minlength=C)).float()
        new = _sweep_by_sorting(hx.gather(lab)[hx.src_idx], dstl, lab, C, cap, size)
        movers = torch.bincount(new[new != lab], minlength=C)
        before = comm.all_gather(movers)[: comm.rank].sum(0)
        lab, _ = _admit(new, lab, C, (cap - size).clamp(min=0), before=before)
    pos = torch.arange(C, device=dev)
    if arrange and C > 2:
        q = comm.all_reduce(torch.bincount(lab[dstl] * C + hx.gather(lab)[hx.src_idx], minlength=C * C))
        if comm.rank == 0:      # one rank arranges, everyone gets ITS answer (eigenvectors are not bit-stable across devices)
            pos = arrange_communities(q.view(C, C).double(),
                                      device=dev if (C > 1500 and dev.type == "cuda") else None).to(dev)
        pos = comm.broadcast(pos.contiguous())
    # new id = (nodes in communities arranged before mine) + (members of my community on lower ranks) + (my index in it)
    counts = comm.all_gather(torch.bincount(lab, minlength=C))              # [world, C]
    size = counts.sum(0)
    by_pos = torch.argsort(pos)
    start = torch.empty(C, dtype=torch.int64, device=dev)
    start[by_pos] = torch.cumsum(size[by_pos], 0) - size[by_pos]
    n = hi - lo
    ar = torch.arange(n, device=dev)
    o = torch.argsort(lab * max(n, 1) + ar)
    sl = lab[o]
    first = torch.searchsorted(sl, torch.arange(C, device=dev))
    new_id = torch.empty(n, dtype=torch.int64, device=dev)
    new_id[o] = start[sl] + counts[: comm.rank].sum(0)[sl] + (ar - first[sl])
    return new_id, lab


def relabel_edges(edge_index, rank):
    """edge_index with node u renamed rank[u] (edge order kept)."""
    return torch.stack([rank[edge_index[0]], rank[edge_index[1]]])


def halo_stats(edge_index, num_nodes, parts):
    """(max halo rows per rank, local-source share of the edges) of the balanced 1-D partition."""
    from .dist import balanced_bounds

    src, dst = edge_index[0], edge_index[1]
    b = balanced_bounds(dst, num_nodes, parts)
    halo, loc, tot = [], 0, 0
    for r in range(parts):
        lo, hi = b[r], b[r + 1]
        m = (dst >= lo) & (dst < hi)
        s = src[m]
        rem = s[(s < lo) | (s >= hi)]
        halo.append(int(torch.unique(rem).numel()))
        loc += int(m.sum()) - int(rem.numel())
        tot += int(m.sum())
    return max(halo), loc / max(tot, 1)
