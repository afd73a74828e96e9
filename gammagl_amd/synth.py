"""Seeded synthetic graphs with the node / edge counts of the benchmark datasets.

No dataset can be downloaded here, so the bench uses R-MAT graphs ((a,b,c,d) = (0.57,0.19,0.19,0.05))
of matching |V| and |E|: symmetrised, de-duplicated to the target directed edge count, node ids
randomly relabelled (worst-case gather locality), edges ordered by source — the layout a coalesced
``edge_index`` has in GammaGL/PyG — and self-loops appended at the end exactly as
``add_self_loops`` does before training (examples/gcn/gcn_trainer.py:58).  Runs on whatever device
it is given (GPU for the big ones).
"""
import math

import torch

DATASETS = {  # name: (nodes, directed edges without loops, input features, classes)
    "cora": (2708, 10556, 1433, 7),
    "arxiv": (169343, 2315598, 128, 40),
    "reddit": (232965, 114615892, 602, 41),
    "products": (2449029, 123718280, 100, 47),
    "papers100M": (111059956, 3231371744, 128, 172),   # symmetrised edge count; needs >= 4 GPUs
}


def rmat_pairs(scale, m, gen, device, a=0.57, b=0.19, c=0.19):
    u = torch.zeros(m, dtype=torch.int64, device=device)
    v = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, generator=gen, device=device)
        ubit = (r >= a + b).to(torch.int64)                      # quadrants c, d
        vbit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)  # quadrants b, d
        u = (u << 1) | ubit
        v = (v << 1) | vbit
    return u, v


def rmat_graph(num_nodes, num_directed_edges, seed=0, device="cpu", relabel="random",
               order="src", self_loops=True):
    """Returns edge_index [2, E] int64 (E = num_directed_edges (+ num_nodes loops))."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    N = int(num_nodes)
    target = int(num_directed_edges) // 2           # undirected pairs
    scale = max(1, math.ceil(math.log2(max(N, 2))))
    keys = torch.empty(0, dtype=torch.int64, device=device)
    rounds = 0
    while keys.numel() < target:
        need = target - keys.numel()
        m = int(need * 1.5) + 1024
        u, v = rmat_pairs(scale, m, gen, device)
        ok = (u < N) & (v < N) & (u != v)
        u, v = u[ok], v[ok]
        lo, hi = torch.minimum(u, v), torch.maximum(u, v)
        keys = torch.unique(torch.cat([keys, lo * N + hi]))
        del u, v, lo, hi, ok
        rounds += 1
        if rounds > 64:
            raise RuntimeError("R-MAT generator cannot reach the requested edge count")
    if keys.numel() > target:
        sel = torch.randperm(keys.numel(), generator=gen, device=device)[:target]
        keys = keys[sel]
    lo, hi = keys // N, keys % N
    del keys
    if relabel == "random":
        pi = torch.randperm(N, generator=gen, device=device)
        lo, hi = pi[lo], pi[hi]
    src = torch.cat([lo, hi])
    dst = torch.cat([hi, lo])
    del lo, hi
    if relabel == "degree":  # hubs first: the locality-friendly ordering
        deg = torch.bincount(dst, minlength=N)
        rank = torch.empty(N, dtype=torch.int64, device=device)
        rank[torch.argsort(deg, descending=True, stable=True)] = torch.arange(N, device=device)
        src, dst = rank[src], rank[dst]
    key = (src * N + dst) if order == "src" else (dst * N + src)
    o = torch.argsort(key)
    src, dst = src[o], dst[o]
    del key, o
    if self_loops:
        loops = torch.arange(N, dtype=torch.int64, device=device)
        src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
    return torch.stack([src, dst]).contiguous()


def dataset_like(name, seed=0, device="cpu", **kw):
    n, e, f, c = DATASETS[name]
    return rmat_graph(n, e, seed=seed, device=device, **kw), n, f, c


def homophilous_graph(n, f, c, deg=2, p_same=0.85, signal=0.5, seed=0, device="cpu"):
    """Seeded node-classification toy with learnable structure (no dataset can be downloaded here): labels y,
    features = noise + `signal` * one_hot(y), `deg` out-edges per node of which `p_same` stay inside the
    node's class, symmetrised.  Returns x [n,f], y [n], edge_index [2, 2*deg*n] (no self-loops)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    y = torch.randint(0, c, (n,), generator=g, device=dev)
    src = torch.arange(n, device=dev).repeat_interleave(deg)
    same = torch.rand(src.shape[0], generator=g, device=dev) < p_same
    order = torch.argsort(y * n + torch.arange(n, device=dev))           # nodes grouped by class
    start = torch.searchsorted(y[order].contiguous(), torch.arange(c + 1, device=dev))
    r = torch.rand(src.shape[0], generator=g, device=dev)
    in_class = order[(start[y[src]] + (r * (start[y[src] + 1] - start[y[src]])).long()).clamp(max=n - 1)]
    anywhere = torch.randint(0, n, (src.shape[0],), generator=g, device=dev)
    dst = torch.where(same, in_class, anywhere)
    ei = torch.cat([torch.stack([src, dst]), torch.stack([dst, src])], dim=1)
    x = torch.randn(n, f, generator=g, device=dev) + signal * torch.nn.functional.one_hot(y, f).float()
    return x, y, ei


# ---------------------------------------------------------------------------------------------------
# World-size-independent, per-rank construction of the same R-MAT graph
# ---------------------------------------------------------------------------------------------------
# `rmat_graph` above draws from torch's sequential generator, so only ONE process can build the graph
# and everybody else has to receive the whole [2, E] edge list.  At the papers100M size (E = 3.23 G,
# 52 GB of int64) no rank may hold that.  `rmat_partitioned` is counter-based instead: candidate pair
# i of the stream is a pure function of (seed, i), so rank r of P generates only its slice of the
# stream, routes every directed edge to the rank that owns it, and the ranks agree on the result
# through a handful of small reductions.  The graph (edge set, node order, ownership bounds) is the
# SAME for every world size — a world-2 run holds exactly the halves of the world-1 graph
# (tests/test_dist_gloo.py) — and no rank ever holds more than ~its share of the edges.

_U64 = (1 << 64) - 1


def _s64(c):
    """Python int (mod 2^64) -> the signed value torch's int64 arithmetic wraps to."""
    c &= _U64
    return c - (1 << 64) if c >= (1 << 63) else c


def _mix64_int(x):
    """splitmix64 finaliser on a Python int (a bijection of 64-bit words)."""
    x &= _U64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _U64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _U64
    return x ^ (x >> 31)


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64(x):
    """The same finaliser on an int64 tensor (wrap-around arithmetic; logical shifts spelled out)."""
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def rmat_pairs_ctr(scale, idx, seed, a=0.57, b=0.19, c=0.19):
    """R-MAT endpoints of the candidate pairs with stream positions `idx` (int64 tensor): quadrant of
    level l from 32 bits of mix64(idx ^ salt(seed, l // 2)) — no generator state, any slice on any rank."""
    u = torch.zeros_like(idx)
    v = torch.zeros_like(idx)
    t_a, t_ab, t_abc = int(a * 2**32), int((a + b) * 2**32), int((a + b + c) * 2**32)
    h = None
    for lvl in range(scale):
        if lvl % 2 == 0:
            h = mix64(idx ^ _s64(_mix64_int(seed * 0x9E3779B97F4A7C15 + 2 * (lvl // 2) + 1)))
            t = _lsr(h, 32)
        else:
            t = h & 0xFFFFFFFF
        ubit = (t >= t_ab).to(torch.int64)
        vbit = (((t >= t_a) & (t < t_ab)) | (t >= t_abc)).to(torch.int64)
        u = (u << 1) | ubit
        v = (v << 1) | vbit
    return u, v


class _Comm:
    """The few collectives the builder needs; world == 1 needs no process group."""

    def __init__(self, rank, world, group, always=False):
        """`always` (tests): go through torch.distributed even with one rank, so that the collectives' dtype /
        split handling is exercised on the real backend (RCCL) by a single-GPU box."""
        self.rank, self.world, self.group = int(rank), int(world), group
        self.always = bool(always)

    def all_reduce(self, t, op="sum"):
        if self.world > 1 or self.always:
            import torch.distributed as dist

            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)
        return t

    def route(self, payload, owner):
        """Send row i of `payload` ([m] or [m, c] int64) to rank owner[i]; returns what this rank receives."""
        if self.world == 1 and not self.always:
            return payload
        import torch.distributed as dist

        order = torch.argsort(owner, stable=True)
        send = payload[order].contiguous()
        sc = torch.bincount(owner, minlength=self.world)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        out = torch.empty((int(rc.sum()),) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
        dist.all_to_all_single(out, send, rc.tolist(), sc.tolist(), group=self.group)
        return out

    def all_to_all_counts(self, sc):
        """What every rank sends me, given what I send every rank (`sc` int64 [world])."""
        if self.world == 1 and not self.always:
            return sc
        import torch.distributed as dist

        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        return rc

    def all_gather(self, t):
        """[world, *t.shape]: the same-shaped tensor of every rank."""
        if self.world == 1 and not self.always:
            return t.unsqueeze(0)
        import torch.distributed as dist

        bufs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(bufs, t.contiguous(), group=self.group)
        return torch.stack(bufs)

    def broadcast(self, t, src=0):
        if self.world > 1 or self.always:
            import torch.distributed as dist

            dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)
        return t

    def all_gather_var(self, t):
        """Concatenation over ranks of 1-D tensors of different lengths (small ones only)."""
        if self.world == 1 and not self.always:
            return t
        import torch.distributed as dist

        n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
        ns = [torch.empty_like(n) for _ in range(self.world)]
        dist.all_gather(ns, n, group=self.group)
        ns = [int(x) for x in ns]
        cap = max(max(ns), 1)
        buf = torch.zeros(cap, dtype=t.dtype, device=t.device)
        buf[: t.numel()] = t
        bufs = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(bufs, buf, group=self.group)
        return torch.cat([b[:k] for b, k in zip(bufs, ns)])


def bounds_from_degree(deg, world):
    """Contiguous node ranges with ~equal numbers of in-edges, from the global in-degree vector
    (identical on every rank; the distributed counterpart of dist.balanced_bounds)."""
    n = int(deg.shape[0])
    cum = torch.cumsum(deg, 0)
    total = int(cum[-1]) if n > 0 else 0
    targets = torch.arange(1, world, device=deg.device, dtype=torch.float64) * (total / world)
    cuts = torch.searchsorted(cum.double(), targets).clamp(max=n)
    return [0] + [int(c) for c in cuts.tolist()] + [n]


def rmat_partitioned(num_nodes, num_directed_edges, seed=0, rank=0, world=1, group=None, device="cpu",
                     relabel="random", order="src", parts=None, stats=None, buckets=None, slab=1 << 27,
                     _always_comm=False):
    """This rank's share of the R-MAT graph with `num_nodes` nodes and exactly `num_directed_edges`
    directed edges (symmetrised, de-duplicated) + one self-loop per node, for a 1-D partition into `parts`
    (default: `world`) contiguous node ranges balanced by in-edge count.

    Returns a dict: src (global ids) / dst (LOCAL ids, dst - lo) / w (symmetric GCN norm
    deg^-1/2[src] * deg^-1/2[dst], degrees counted on the looped edge list like calc_gcn_norm) of the
    edges whose destination this rank owns; bounds [parts+1]; e_global (Python int, may exceed 2^31);
    deg (global in-degree incl. the loop, float32 [N]).  With world == 1 and parts == P > 1 the process
    plays rank `rank` of a P-way partition (single-GPU share probes, dry partitions).

    Memory: candidates are generated in slabs of `slab` stream positions, and the directed edges a process
    holds are kept in `buckets` hash buckets (1 per rank in a distributed run; with world == 1 as many as
    keeps a bucket under 2^29 keys), so no sort / unique ever runs on more than a bucket — a single GPU can
    build the papers100M-sized graph (3.2 G edges) piecewise to cut one rank's share out of it."""
    dev = torch.device(device)
    N, P = int(num_nodes), int(world)
    parts = int(parts or world)
    comm = _Comm(rank if P > 1 else 0, P, group, _always_comm)   # (world == 1, parts > 1: `rank` only picks the share played)
    T = int(num_directed_edges) // 2
    if buckets is None:
        buckets = 1 if P > 1 else max(1, -(-int(2.6 * T) // (1 << 29)))
    nb = int(buckets)
    scale = max(1, math.ceil(math.log2(max(N, 2))))
    peak = 0
    pi = None
    if relabel == "random":
        # the same permutation on every rank (CPU generator: identical whatever the device)
        pi = torch.randperm(N, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
    # directed edges held here, as keys dst * N + src, in nb hash buckets of the destination (every copy of a
    # directed edge meets in one bucket of one rank, so the de-duplication there is complete; hashing keeps
    # the provisional load balanced whatever the node order)
    keys = [torch.empty(0, dtype=torch.int64, device=dev) for _ in range(nb)]

    def bucket_id(dk):
        return _lsr(mix64(dk // N), 33) % (P * nb)

    U, base, rounds = 0, 0, 0
    while U < T:
        m = int((T - U) * 1.5) + 1024                        # candidates this round, over all ranks
        i0, i1 = base + m * comm.rank // P, base + m * (comm.rank + 1) // P
        pending = [[] for _ in range(nb)]
        for s0 in range(i0, max(i1, i0 + 1), slab):
            idx = torch.arange(s0, min(s0 + slab, i1), dtype=torch.int64, device=dev)
            u, v = rmat_pairs_ctr(scale, idx, seed)
            del idx
            ok = (u < N) & (v < N) & (u != v)
            u, v = u[ok], v[ok]
            if pi is not None:
                u, v = pi[u], pi[v]
            dkey = torch.cat([v * N + u, u * N + v])         # both directions of every pair
            del u, v, ok
            if P > 1 or _always_comm:
                recv = comm.route(dkey, bucket_id(dkey) // nb)
                peak = max(peak, int(dkey.numel()) + int(recv.numel()))
                dkey = recv
            if nb == 1:
                pending[0].append(dkey)
            else:
                b = bucket_id(dkey) % nb
                o = torch.argsort(b)
                dkey, cnt = dkey[o], torch.bincount(b, minlength=nb).tolist()
                del b, o
                for j, piece in enumerate(torch.split(dkey, cnt)):
                    pending[j].append(piece.clone())
            del dkey
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        for j in range(nb):
            keys[j] = torch.unique(torch.cat([keys[j]] + pending[j]))
            pending[j] = None
            cnt += ((keys[j] % N) < (keys[j] // N)).sum()    # canonical (src < dst) copies = undirected pairs
        peak = max(peak, sum(int(k.numel()) for k in keys))
        U = int(comm.all_reduce(cnt))
        base += m
        rounds += 1
        if rounds > 64:
            raise RuntimeError("R-MAT generator cannot reach the requested edge count")
    salt = _s64(_mix64_int(seed + 0x51ED27))

    def pair_hash(k):
        sr, ds = k % N, k // N
        return mix64((torch.minimum(sr, ds) * N + torch.maximum(sr, ds)) ^ salt)

    if U > T:
        # keep the T pairs with the smallest hash of their canonical key (signed order; mix64 is a
        # bijection, so there are no ties): a selection that does not depend on who holds what
        hist = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
        for k in keys:
            h = pair_hash(k)
            hist += torch.bincount(((h >> 44) + (1 << 19))[(k % N) < (k // N)], minlength=1 << 20)
        hist = comm.all_reduce(hist)
        cum = torch.cumsum(hist, 0)
        b = int(torch.searchsorted(cum, torch.tensor([T], device=dev, dtype=cum.dtype)))
        below = int(cum[b - 1]) if b > 0 else 0
        cand = []
        for k in keys:
            h = pair_hash(k)
            cand.append(h[((k % N) < (k // N)) & (((h >> 44) + (1 << 19)) == b)])
        cand = comm.all_gather_var(torch.cat(cand))
        thr = torch.sort(cand).values[T - below - 1]
        keys = [k[pair_hash(k) <= thr] for k in keys]
    # global in-degree (the graph is symmetric: also the out-degree), without the loops yet
    deg = torch.zeros(N, dtype=torch.int64, device=dev)
    for k in keys:
        deg += torch.bincount(k // N, minlength=N)
    deg = comm.all_reduce(deg)
    rk = None
    if relabel == "degree":   # hubs first: the locality-friendly ordering
        rk = torch.empty(N, dtype=torch.int64, device=dev)
        o = torch.argsort(deg, descending=True, stable=True)
        rk[o] = torch.arange(N, device=dev)
        deg = deg[o]
        del o
    deg = deg + 1                                             # add_self_loops: one loop per node
    bounds = bounds_from_degree(deg, parts)
    bt = torch.tensor(bounds[1:-1], device=dev, dtype=torch.int64)
    me = comm.rank if P > 1 else int(rank)
    lo_n, hi_n = bounds[me], bounds[me + 1]
    e_total = sum(int(k.numel()) for k in keys)
    srcs, dsts, send = [], [], None
    dry = P == 1 and parts > 1
    if dry:
        send = [[] for _ in range(parts)]
    for j in range(nb):
        k = keys[j]
        keys[j] = None
        sr, ds = k % N, k // N
        del k
        if rk is not None:
            sr, ds = rk[sr], rk[ds]
        if P > 1 or (_always_comm and parts == 1):
            own = torch.searchsorted(bt, ds, right=True)
            got = comm.route(torch.stack([sr, ds], 1), own)
            peak = max(peak, int(sr.numel()) + int(got.shape[0]))
            srcs.append(got[:, 0].contiguous())
            dsts.append(got[:, 1].contiguous())
            del got, own
        else:
            mine = (ds >= lo_n) & (ds < hi_n)
            srcs.append(sr[mine])
            dsts.append(ds[mine])
            if dry:   # which of my rows the other parts need (what they would request in a real P-rank run)
                out = (sr >= lo_n) & (sr < hi_n) & ~mine
                own = torch.searchsorted(bt, ds[out], right=True)
                so = sr[out]
                for q in range(parts):
                    if q != me:
                        send[q].append(torch.unique(so[own == q]))
                del out, own, so
            del mine
        del sr, ds
    src, dst = torch.cat(srcs), torch.cat(dsts)
    del srcs, dsts
    key = (src * N + dst) if order == "src" else (dst * N + src)
    o = torch.argsort(key)
    src, dst = src[o], dst[o]
    del key, o
    loops = torch.arange(lo_n, hi_n, dtype=torch.int64, device=dev)
    src, dst = torch.cat([src, loops]), torch.cat([dst, loops])   # loops appended last, as add_self_loops does
    dis = deg.to(torch.float32).pow(-0.5)
    w = (dis[src] * dis[dst]).contiguous()
    e_loc = torch.tensor([src.numel()], dtype=torch.int64, device=dev)
    out = {"src": src.contiguous(), "dst": (dst - lo_n).contiguous(), "w": w, "bounds": bounds,
           "deg": deg.to(torch.float32), "num_nodes": N, "rank": me, "parts": parts}
    if P > 1 or parts == 1:
        out["e_global"] = int(comm.all_reduce(e_loc))
    else:
        out["e_global"] = e_total + N
        out["send_rows"] = [(torch.unique(torch.cat(t)) - lo_n) if t else torch.empty(0, dtype=torch.int64, device=dev)
                            for t in send]
    peak = max(peak, int(src.numel()))
    if stats is not None:
        stats.update(peak_edges=peak, rounds=rounds, local_edges=int(src.numel()), buckets=nb)
    return out


# ---------------------------------------------------------------------------------------------------
# Graphs every rank can hold (products-sized and below): planted communities, locality-aware orders
# ---------------------------------------------------------------------------------------------------
def cut_share(src, dst, num_nodes, parts=1, rank=0, order="src", dry=False):
    """This rank's share (same dict as `rmat_partitioned`) of a graph whose FULL directed edge list
    (`src`, `dst`: global ids in their final labelling, no self-loops) the caller holds — the route for
    graphs that fit one GPU (2 GB of ids at the products size), where every rank can build the whole list
    itself and no construction-time collective is needed.  Bounds, loops, symmetric GCN norm and edge order
    exactly as `rmat_partitioned` produces them; `dry`: also the send lists the other parts would request."""
    dev = src.device
    N, parts, me = int(num_nodes), int(parts), int(rank)
    deg = torch.bincount(dst, minlength=N) + 1              # + the self-loop (add_self_loops)
    bounds = bounds_from_degree(deg, parts)
    lo, hi = bounds[me], bounds[me + 1]
    mine = (dst >= lo) & (dst < hi)
    s, d = src[mine], dst[mine]
    key = (s * N + d) if order == "src" else (d * N + s)
    o = torch.argsort(key)
    s, d = s[o], d[o]
    del key, o
    loops = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    s, d = torch.cat([s, loops]), torch.cat([d, loops])
    dis = deg.to(torch.float32).pow(-0.5)
    out = {"src": s.contiguous(), "dst": (d - lo).contiguous(), "w": (dis[s] * dis[d]).contiguous(),
           "bounds": bounds, "deg": deg.to(torch.float32), "num_nodes": N, "rank": me, "parts": parts,
           "e_global": int(src.numel()) + N}
    if dry:
        bt = torch.tensor(bounds[1:-1], device=dev, dtype=torch.int64)
        outward = (src >= lo) & (src < hi) & ~mine
        own = torch.searchsorted(bt, dst[outward], right=True)
        so = src[outward]
        out["send_rows"] = [(torch.unique(so[own == q]) - lo) if q != me else torch.empty(0, dtype=torch.int64, device=dev)
                            for q in range(parts)]
    return out


def repartition(g, new_id, comm, order="src"):
    """A rank's share `g` (the dict `rmat_partitioned` returns, world == parts) after the nodes are RENAMED: `new_id`
    int64 [n_local] = new global id of every node this rank owns (a permutation of 0..N-1 over the ranks, e.g.
    `partition.cluster_order_distributed`).  New balanced bounds are cut from the renamed in-degrees, every edge is
    routed to the rank that owns its renamed destination, order / loops / symmetric norm as `rmat_partitioned` leaves
    them.  No rank ever holds more than its share (+ the [N] degree vector the shares carry anyway): the relabelling
    step for graphs that do not fit one GPU.  The result is what `cut_share` makes of the renamed full edge list
    (tests/test_dist_gloo.py checks exactly that)."""
    from .partition import HaloIndex

    src, dstl, bounds, N = g["src"], g["dst"], g["bounds"], int(g["num_nodes"])
    dev = src.device
    lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
    hx = HaloIndex(src, lo, hi, bounds, comm)
    ns = hx.gather(new_id)[hx.src_idx]                       # renamed sources (halo ids fetched from their owners)
    nd = new_id[dstl]
    keep = ns != nd                                          # the loops are re-made for the new ranges
    ns, nd = ns[keep], nd[keep]
    deg = comm.all_reduce(torch.bincount(nd, minlength=N)) + 1
    nb = bounds_from_degree(deg, comm.world)
    bt = torch.tensor(nb[1:-1], device=dev, dtype=torch.int64)
    got = comm.route(torch.stack([ns, nd], 1), torch.searchsorted(bt, nd, right=True))
    s, d = got[:, 0].contiguous(), got[:, 1].contiguous()
    del got, ns, nd
    o = torch.argsort((s * N + d) if order == "src" else (d * N + s))
    s, d = s[o], d[o]
    lo, hi = nb[comm.rank], nb[comm.rank + 1]
    loops = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    s, d = torch.cat([s, loops]), torch.cat([d, loops])
    dis = deg.to(torch.float32).pow(-0.5)
    e_loc = torch.tensor([s.numel()], dtype=torch.int64, device=dev)
    return {"src": s.contiguous(), "dst": (d - lo).contiguous(), "w": (dis[s] * dis[d]).contiguous(), "bounds": nb,
            "deg": deg.to(torch.float32), "num_nodes": N, "rank": comm.rank, "parts": comm.world,
            "e_global": int(comm.all_reduce(e_loc))}


PLANTED_LEVELS = ((1024, 0.60), (64, 0.25), (8, 0.10))   # (groups, share of a node's edges that stay inside its group)


def planted_pairs(num_nodes, out_deg=25, levels=PLANTED_LEVELS, seed=0, device="cpu", slab=1 << 26):
    """Directed edge list (no loops, symmetrised, de-duplicated) of a HIERARCHICAL planted-community graph in its
    natural labelling: `levels` = ((G1, p1), (G2, p2), ...) from the finest grouping to the coarsest, every grouping
    a partition of the ids into G contiguous ranges (group of node u = u G // N; each G a multiple of the next, so
    the groupings nest).  Node u draws `out_deg` neighbours: with probability p1 inside its finest group, p2 inside
    its next-coarser group, ..., anywhere with what is left.  Counter-based (edge j of node u is a pure function of
    (seed, u, j)): every rank builds the same list.  Stands in for the locality a co-purchase / citation graph has
    at every scale and R-MAT lacks ("locality-aware partition"); the default has communities of
    ~N / 1024 nodes (2 400 at the products size: 0.6 MB of 64-column feature slices, L2-sized)."""
    dev = torch.device(device)
    N = int(num_nodes)
    cum, acc = [], 0.0
    for _, pr in levels:
        acc += float(pr)
        cum.append(int(min(acc, 1.0) * 2**32))
    salt1 = _s64(_mix64_int(seed * 0x9E3779B97F4A7C15 + 0xA1))
    salt2 = _s64(_mix64_int(seed * 0x9E3779B97F4A7C15 + 0xB2))
    keys = []
    total = N * int(out_deg)
    for s0 in range(0, total, slab):
        idx = torch.arange(s0, min(s0 + slab, total), dtype=torch.int64, device=dev)
        u = idx // int(out_deg)
        h1, h2 = mix64(idx ^ salt1), mix64(idx ^ salt2)
        t = _lsr(h1, 32)
        lo, hi = torch.zeros_like(u), torch.full_like(u, N)          # "anywhere" unless a level claims the draw
        for (G, _), thr in reversed(list(zip(levels, cum))):          # coarsest first, finer levels overwrite
            grp = (u * int(G)) // N
            g_lo, g_hi = (grp * N + G - 1) // G, ((grp + 1) * N + G - 1) // G
            take = t < thr
            lo, hi = torch.where(take, g_lo, lo), torch.where(take, g_hi, hi)
        v = lo + _lsr(h2, 1) % (hi - lo).clamp(min=1)
        ok = u != v
        u, v = u[ok], v[ok]
        keys.append(torch.unique(torch.minimum(u, v) * N + torch.maximum(u, v)))
        del idx, u, v, h1, h2, t, lo, hi, ok
    k = torch.unique(torch.cat(keys)) if len(keys) > 1 else keys[0]
    a, b = k // N, k % N
    return torch.cat([a, b]), torch.cat([b, a])


def full_graph_partitioned(kind, num_nodes, num_directed_edges, seed=0, rank=0, world=1, device="cpu",
                           relabel="random", order="src", parts=None, stats=None, eng=None, clusters=None):
    """`rmat_partitioned`'s result for graphs every rank can build whole: `kind` = "rmat" (the same graph
    `rmat_partitioned` makes — used when the requested order needs the whole graph) or "planted"
    (`planted_pairs`, ~`num_directed_edges` edges).  `relabel`: "random" | "none" | "degree" |
    "cluster" (`partition.cluster_order` on the randomly relabelled graph: what a user would run on a graph
    whose ids carry no locality).  No collectives: every rank computes the same labelling."""
    dev = torch.device(device)
    N = int(num_nodes)
    parts = int(parts or world)
    me = int(rank)
    if kind == "rmat":
        g = rmat_partitioned(N, num_directed_edges, seed=seed, device=dev, relabel="random" if relabel == "cluster" else relabel,
                             order="src")
        src, dst = g["src"][:-N], g["dst"][:-N]              # (world = 1: dst is global; the loops come last)
        del g
    elif kind == "planted":
        out_deg = max(2, int(num_directed_edges) // (2 * N))
        src, dst = planted_pairs(N, out_deg=out_deg, seed=seed, device=dev)
        if relabel != "none":
            pi = torch.randperm(N, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
            src, dst = pi[src], pi[dst]
    else:
        raise ValueError(kind)
    if relabel == "degree" and kind != "rmat":
        deg = torch.bincount(dst, minlength=N)
        rk = torch.empty(N, dtype=torch.int64, device=dev)
        rk[torch.argsort(deg, descending=True, stable=True)] = torch.arange(N, device=dev)
        src, dst = rk[src], rk[dst]
    if relabel == "cluster":
        from .partition import cluster_order

        ei = torch.stack([src, dst]).contiguous()
        if clusters is None:
            # several times more labels than there can be communities worth finding: small labels stay pure and the
            # propagation merges them (products size: 4 081 labels -> ~1 400 communities of purity 0.998 against the
            # planted 2 400-node groups in 30 sweeps, 1.7 s; 1 020 labels reach 0.70, profiles/r3_cluster_quality.txt)
            clusters = max(8, min(8192, N // 600))
        rk, lab = cluster_order(ei, N, clusters=clusters, sweeps=30, seed=seed, eng=eng)
        if eng is not None:
            eng.clear_caches()
        del ei
        src, dst = rk[src], rk[dst]
        if stats is not None:
            stats["clusters"] = int(lab.max()) + 1
    out = cut_share(src, dst, N, parts=parts, rank=me, order=order, dry=(world == 1 and parts > 1))
    if stats is not None:
        stats.update(peak_edges=int(src.numel()), local_edges=int(out["src"].numel()), rounds=1, buckets=1)
    return out
