"""Checks of ONE rank's share of a P-way partition built by the per-rank generator and run as a dry partition
(shared by the -m gpu config-5 test at the papers100M size and a CPU test at toy size on the host-emulated kernels)."""
import torch


def check_rank_share(eng, dev, n, e, P, r, min_buckets=1, buckets=None, mem_limit=None):
    from gammagl_amd import dist as gd
    from gammagl_amd.dist import PartitionedGraph
    from gammagl_amd.synth import rmat_partitioned

    stats = {}
    g = rmat_partitioned(n, e, seed=0, rank=r, world=1, device=dev, parts=P, stats=stats, buckets=buckets)
    bounds, deg = g["bounds"], g["deg"]
    lo, hi = bounds[r], bounds[r + 1]
    # 64-bit bookkeeping: the global edge count is past 2^31 and exact; a rank's own share stays below it (int32 perm)
    assert g["e_global"] == e + n and isinstance(g["e_global"], int)
    e_local = int(g["src"].numel())
    assert 0.8 * g["e_global"] / P < e_local < 1.25 * g["e_global"] / P and e_local < 2**31
    assert bounds[0] == 0 and bounds[-1] == n and all(a < b for a, b in zip(bounds[:-1], bounds[1:]))
    assert int(deg[lo:hi].double().sum()) == e_local          # in-degrees (loops incl.) of the owned rows = the share
    assert stats["buckets"] >= min_buckets and stats["peak_edges"] < 2.8 * g["e_global"]   # built piecewise, never [2, E] at once
    pg = PartitionedGraph.from_local(g["src"], g["dst"], g["w"], bounds, n, g["e_global"], rank=r, world=1, eng=eng,
                                     send_rows=g["send_rows"])
    del g
    assert pg.dry and pg.n_local == hi - lo and pg.e_local == e_local and pg.gp_loc.E + pg.gp_halo.E == e_local
    for plan in (pg.gp_loc.fwd, pg.gp_halo.fwd):
        assert plan.rowptr.dtype == torch.int64 and int(plan.rowptr[-1]) == plan.E and plan.N == pg.n_local
    # halo identities
    hid = pg.halo_ids
    assert pg.n_halo == hid.numel() > 0 and bool((hid[1:] > hid[:-1]).all()) and int(hid[0]) >= 0 and int(hid[-1]) < n
    assert not bool(((hid >= lo) & (hid < hi)).any())
    assert sum(pg.recv_splits) == pg.n_halo and pg.recv_splits[r] == 0 and len(pg.recv_splits) == P
    bt = torch.tensor(bounds, device=dev)
    assert pg.recv_splits == (torch.searchsorted(hid, bt)[1:] - torch.searchsorted(hid, bt)[:-1]).tolist()
    hs = pg.ei_halo[0]
    seen = torch.zeros(pg.n_halo, dtype=torch.bool, device=dev)
    seen[hs] = True
    assert int(hs.max()) < pg.n_halo and bool(seen.all())      # every halo row is read by some edge
    del seen
    # send lists: inside my range, sorted + distinct per peer, and — the graph being symmetric — the rows peer q needs
    # from me are exactly my rows with an in-edge from q's range
    assert len(pg.send_splits) == P and pg.send_splits[r] == 0 and sum(pg.send_splits) == pg.n_send
    owner = torch.searchsorted(bt[1:-1].contiguous(), hid[hs], right=True)
    off = 0
    for q in range(P):
        seg = pg.send_idx[off:off + pg.send_splits[q]]
        off += pg.send_splits[q]
        if q == r:
            continue
        assert bool((seg[1:] > seg[:-1]).all()) and int(seg[0]) >= 0 and int(seg[-1]) < pg.n_local
        mirror = torch.unique(pg.ei_halo[1][owner == q])
        assert torch.equal(seg, mirror), q
    del owner, hs
    # the share's aggregate with a filled halo buffer: f64 column checksum + f64 row evaluations
    K = 64
    gen = torch.Generator(device=dev).manual_seed(1)
    h = torch.randn(pg.n_local, K, generator=gen, device=dev)
    halo = torch.randn(pg.n_halo, K, generator=gen, device=dev)

    class _Filled:
        def wait(self):
            return True

    def fill(out_rows, inp, out_splits, in_splits, tag="a2a"):        # what the 7 peers would have sent
        assert out_rows == pg.n_halo and inp.shape[0] == pg.n_send
        return halo, _Filled()

    pg._a2a = fill
    assert len(gd._HaloAggregate._chunks(K)) == 1
    out = pg.aggregate(h)
    chk = torch.zeros(8, dtype=torch.float64, device=dev)
    for ei, w, x in ((pg.ei_loc, pg.w_loc, h), (pg.ei_halo, pg.w_halo, halo)):
        for s in range(0, ei.shape[1], 64_000_000):
            sl = slice(s, min(ei.shape[1], s + 64_000_000))
            chk += (w[sl].double().unsqueeze(1) * x[ei[0, sl], :8].double()).sum(0)
    got = out[:, :8].double().sum(0)
    assert float(((got - chk).abs() / chk.abs().clamp(min=1.0)).max()) < 1e-6
    rows = torch.randint(0, pg.n_local, (24,), generator=gen, device=dev).tolist()
    rows.append(int(torch.argmax(pg.gp_halo.fwd.counts())))            # the heaviest row too
    for i in rows:
        want = torch.zeros(K, dtype=torch.float64, device=dev)
        for gp, w, x in ((pg.gp_loc, pg.w_loc, h), (pg.gp_halo, pg.w_halo, halo)):
            b, e_ = int(gp.fwd.rowptr[i]), int(gp.fwd.rowptr[i + 1])
            pos = torch.arange(b, e_, device=dev)
            orig = gp.fwd.perm[pos].long() if gp.fwd.perm is not None else pos
            want += (w[orig].double().unsqueeze(1) * x[gp.col[pos].long()].double()).sum(0)
        mag = float(want.abs().max()) + 1.0
        assert float((out[i].double() - want).abs().max()) < 2e-5 * mag, i
    # the transposed walks: <A h, g> == <h, A^T g> over local + halo sources (f64)
    hg = h.clone().requires_grad_(True)
    gout = torch.randn(pg.n_local, K, generator=gen, device=dev)
    sent = {}

    def fill_bwd(out_rows, inp, out_splits, in_splits, tag="a2a"):
        if out_rows == pg.n_halo:
            return halo, _Filled()
        sent["ghalo"] = inp                                              # gradient rows that would travel back
        return torch.zeros((pg.n_send, inp.shape[1]), device=dev), _Filled()

    pg._a2a = fill_bwd
    y = pg.aggregate(hg)
    y.backward(gout)
    lhs = float((y.detach().double() * gout.double()).sum())
    rhs = float((hg.grad.double() * h.double()).sum()) + float((sent["ghalo"].double() * halo.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), 1.0)
    if mem_limit is not None:
        assert torch.cuda.max_memory_allocated() < mem_limit
    del pg, h, halo, out, hg, y
    eng.clear_caches()
