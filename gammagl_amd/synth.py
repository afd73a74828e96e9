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
