"""N > 1 path on CPU: world_size 2 and 3 over gloo (127.0.0.1), host-emulated kernels injected as
the Engine.  Checks the node partition + halo all-to-all-v against the unpartitioned result:
forward rows, input gradients (through the reverse exchange), weight gradients after the flat
all-reduce and the loss — all within 1e-5 relative (the association differs, not the arithmetic)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, tmp_path, mode="", timeout=300):
    """Start `world` worker processes of this file on a free 127.0.0.1 port and wait for them; one retry on a fresh
    port (the port is picked, released and re-bound by rank 0: another process can grab it in between)."""
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    rcs = None
    for attempt in range(2):
        port = _free_port()
        for r in range(world):
            ok = tmp_path / f"ok{r}"
            if ok.exists():
                ok.unlink()
        procs = [subprocess.Popen([sys.executable, __file__, str(r), str(world), str(port), str(tmp_path)] + ([mode] if mode else []))
                 for r in range(world)]
        try:
            rcs = [p.wait(timeout=timeout) for p in procs]
        except subprocess.TimeoutExpired:
            for p in procs:
                p.kill()
            rcs = [-9] * world
        if rcs == [0] * world:
            break
    assert rcs == [0] * world, rcs
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _emul_engine():
    sys.path.insert(0, REPO)
    from gammagl_amd import _lib
    from gammagl_amd.ops import Engine

    return Engine(_lib.bind(os.path.join(HERE, "emul", "libggl_emul.so")), require_cuda=False)


def _problem(seed=0):
    from gammagl_amd.synth import rmat_graph

    N, F, Hd, C = 300, 12, 16, 5
    ei = rmat_graph(N, 4000, seed=seed, device="cpu")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, F, generator=g)
    y = torch.randint(0, C, (N,), generator=g)
    train = torch.rand(N, generator=g) < 0.5
    return N, F, Hd, C, ei, x, y, train


def _worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _emul_engine()
        from gammagl_amd.dist import DistGCNTrainer, PartitionedGraph

        N, F, Hd, C, ei, x, y, train = _problem()
        w = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(3)) + 0.1
        pg = PartitionedGraph(ei, w, N, rank, world, eng=eng)
        assert sum(pg.recv_splits) == pg.n_halo and sum(pg.send_splits) == pg.n_send
        # forward / backward of one aggregate vs the unpartitioned op
        h = torch.randn(N, Hd, generator=torch.Generator().manual_seed(4))
        go = torch.randn(N, Hd, generator=torch.Generator().manual_seed(5))
        hl = h[pg.lo:pg.hi].clone().requires_grad_(True)
        out = pg.aggregate(hl)
        out.backward(go[pg.lo:pg.hi])
        hf = h.clone().requires_grad_(True)
        full = eng.c_spmm_sum(ei, w, hf)
        full.backward(go)
        torch.testing.assert_close(out.detach(), full.detach()[pg.lo:pg.hi], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(hl.grad, hf.grad[pg.lo:pg.hi], rtol=1e-5, atol=1e-5)
        # the same exchange as grouped point-to-point transfers (GGL_HALO_A2A=p2p: the A/B switch for the first real
        # multi-GPU run): same rows, same gradients
        import gammagl_amd.dist as gdist
        gdist.A2A_MODE = "p2p"
        try:
            hp = h[pg.lo:pg.hi].clone().requires_grad_(True)
            outp = pg.aggregate(hp)
            outp.backward(go[pg.lo:pg.hi])
            assert torch.equal(outp.detach(), out.detach()) and torch.equal(hp.grad, hl.grad)
        finally:
            gdist.A2A_MODE = "a2a"
        # wide features: the exchange is pipelined in 4 (K = 256) / 2 (K = 136) column chunks
        from gammagl_amd.dist import _HaloAggregate
        assert len(_HaloAggregate._chunks(256)) == 4 and len(_HaloAggregate._chunks(136)) == 2
        for K in (256, 136, 47):
            hk = torch.randn(N, K, generator=torch.Generator().manual_seed(K))
            gk = torch.randn(N, K, generator=torch.Generator().manual_seed(K + 1))
            a = hk[pg.lo:pg.hi].clone().requires_grad_(True)
            b = hk.clone().requires_grad_(True)
            ya = pg.aggregate(a)
            ya.backward(gk[pg.lo:pg.hi])
            yb = eng.c_spmm_sum(ei, w, b)
            yb.backward(gk)
            torch.testing.assert_close(ya.detach(), yb.detach()[pg.lo:pg.hi], rtol=1e-5, atol=1e-4)
            torch.testing.assert_close(a.grad, b.grad[pg.lo:pg.hi], rtol=1e-5, atol=1e-4)
        # one training step (dropout off) vs the world-size-1 run of the same code
        n_train = int(train.sum())
        tr = DistGCNTrainer(pg, F, Hd, C, num_layers=3, drop_rate=0.0, seed=7, device="cpu")
        loc = torch.nonzero(train[pg.lo:pg.hi]).reshape(-1)
        loss = tr.step(x[pg.lo:pg.hi].contiguous(), y[pg.lo:pg.hi].contiguous(), loc, n_train)
        lsum = loss.clone().reshape(1)
        dist.all_reduce(lsum)
        pg1 = PartitionedGraph(ei, w, N, 0, 1, eng=eng)
        tr1 = DistGCNTrainer(pg1, F, Hd, C, num_layers=3, drop_rate=0.0, seed=7, device="cpu")
        loss1 = tr1.step(x, y, torch.nonzero(train).reshape(-1), n_train)
        torch.testing.assert_close(lsum[0], loss1, rtol=1e-5, atol=1e-6)
        for p, p1 in zip(tr.net.parameters(), tr1.net.parameters()):
            torch.testing.assert_close(p.grad, p1.grad, rtol=2e-4, atol=1e-6)
            torch.testing.assert_close(p.detach(), p1.detach(), rtol=1e-4, atol=1e-6)
        # (a) the input features' halo rows travel ONCE: [x_local ; x_halo] is cached on the identity of x, the first
        # layer (either association) then exchanges nothing — same loss and gradients as the step that exchanges its
        # layer-1 rows every time, and per step exactly the all-to-alls of layers 2 and 3 (forward + backward each)
        xl, yl = x[pg.lo:pg.hi].contiguous(), y[pg.lo:pg.hi].contiguous()
        xc = pg.with_halo(xl)
        assert pg.with_halo(xl) is xc and tuple(xc.shape) == (pg.n_local + pg.n_halo, F)
        torch.testing.assert_close(xc[pg.n_local:], x[pg.halo_ids])
        agg_pre = pg.aggregate(xc, halo_included=True)
        torch.testing.assert_close(agg_pre, pg.aggregate(xl), rtol=1e-6, atol=1e-6)
        for af in (False, True):
            res = {}
            for const in (True, False):
                t = DistGCNTrainer(pg, F, Hd, C, num_layers=3, drop_rate=0.0, seed=9, device="cpu", aggregate_first=af,
                                   const_input_halo=const)
                pg.profile = {}
                ls = t.step(xl, yl, loc, n_train)
                calls = pg.profile_summary(1)["a2a_calls"]
                pg.profile = None
                res[const] = (ls, [p.grad.clone() for p in t.net.parameters()], calls)
            assert res[True][2] == 4 and res[False][2] == (5 if af else 6), (af, res[True][2], res[False][2])
            torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-6, atol=1e-7)
            for ga, gb in zip(res[True][1], res[False][1]):
                torch.testing.assert_close(ga, gb, rtol=1e-4, atol=1e-6)
        # (c) persistent exchange buffers: the second step runs on the very buffers of the first
        t = DistGCNTrainer(pg, F, Hd, C, num_layers=3, drop_rate=0.0, seed=9, device="cpu")
        t.step(xl, yl, loc, n_train)
        ptrs = {k: b.data_ptr() for k, b in pg._bufs.items()}
        t.step(xl, yl, loc, n_train)
        assert ptrs and ptrs == {k: b.data_ptr() for k, b in pg._bufs.items()}
        # (d) the number of column chunks of the exchange is MEASURED (tune_halo_chunks): every rank ends up with the
        # same counts (they are numbers of collectives), and the training trajectory — dropout draws included — is the
        # one of a trainer that never tuned
        Hw = 128
        eng.reseed(123)
        ta = DistGCNTrainer(pg, F, Hw, C, num_layers=3, drop_rate=0.5, seed=11, device="cpu")
        la = [ta.step(xl, yl, loc, n_train).clone() for _ in range(2)]
        eng.reseed(123)
        tb = DistGCNTrainer(pg, F, Hw, C, num_layers=3, drop_rate=0.5, seed=11, device="cpu")
        tuned = tb.tune_halo_chunks(xl, yl, loc, n_train, candidates=(1, 2), iters=1)
        assert tuned["exchange"] in (1, 2) and tuned["const"] in (1, 2) and len(tuned["ms"]) == 4
        assert pg.halo_chunks == {"exchange": tuned["exchange"], "const": tuned["const"]}
        assert len(_HaloAggregate._chunks(Hw, pg)) == tuned["exchange"] and len(_HaloAggregate._chunks(48, pg)) == 1
        picks = [None] * world
        dist.all_gather_object(picks, (tuned["exchange"], tuned["const"]))
        assert all(pk == picks[0] for pk in picks), picks
        lb = [tb.step(xl, yl, loc, n_train).clone() for _ in range(2)]
        assert all(torch.equal(a_, b_) for a_, b_ in zip(la, lb)), (la, lb)
        for pa, pb in zip(ta.net.parameters(), tb.net.parameters()):
            assert torch.equal(pa.detach(), pb.detach())
        pg.halo_chunks.clear()
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_matches_single_process(world, tmp_path):
    _run_ranks(world, tmp_path)


def test_balanced_bounds_and_world1():
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    eng = _emul_engine()
    from gammagl_amd.dist import PartitionedGraph, balanced_bounds

    N, F, Hd, C, ei, x, y, train = _problem(1)
    b = balanced_bounds(ei[1], N, 4)
    assert b[0] == 0 and b[-1] == N and all(b[i] <= b[i + 1] for i in range(4))
    deg = torch.bincount(ei[1], minlength=N)
    shares = [int(deg[b[i]:b[i + 1]].sum()) for i in range(4)]
    assert max(shares) < 2.0 * (ei.shape[1] / 4) + int(deg.max())
    w = torch.ones(ei.shape[1])
    pg = PartitionedGraph(ei, w, N, 0, 1, eng=eng)
    assert pg.n_halo == 0 and pg.n_send == 0 and pg.n_local == N
    h = torch.randn(N, 8)
    torch.testing.assert_close(pg.aggregate(h), eng.c_spmm_sum(ei, w, h))


@pytest.mark.parametrize("aggregate_first", [False, True])
def test_distgcn_matches_plain_composition_with_padded_classes(aggregate_first):
    """DistGCN (fused epilogue, class columns padded 10 -> 12 inside the last GEMM, side-stream weight
    gradients off on CPU) == Linear -> spmm -> + bias -> ReLU written out in torch, forward and gradients.
    aggregate_first: the first layer (12 -> 16, input narrower than output) computes (A X) W — the same product
    associated the other way round, so it agrees to the rounding of the terms rather than to 1e-5 of each element —
    runs ONE aggregate forward and none backward (its input carries no gradient)."""
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    eng = _emul_engine()
    from gammagl_amd.dist import DistGCN, PartitionedGraph

    N, F, Hd, _, ei, x, y, train = _problem(3)
    C = 10
    w = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(1)) + 0.1
    pg = PartitionedGraph(ei, w, N, 0, 1, eng=eng)
    torch.manual_seed(0)
    net = DistGCN(F, Hd, C, num_layers=3, drop_rate=0.0, aggregate_first=aggregate_first)
    for b in net.bias:
        torch.nn.init.normal_(b)
    out = net(x, pg)
    assert out.shape == (N, C) and net.agg_per_step == (5 if aggregate_first else 6)
    go = torch.randn(N, C, generator=torch.Generator().manual_seed(2))
    out.backward(go)
    net.join()
    got = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    gp = eng.graph_plan(ei, N)
    h = x
    for i in range(3):
        h = eng.spmm(gp, w, h @ net.lin[i].weight.t()) + net.bias[i]
        if i < 2:
            h = torch.relu(h)
    tol = dict(rtol=1e-5, atol=1e-5) if not aggregate_first else dict(rtol=1e-4, atol=1e-4 * float(h.abs().max()))
    torch.testing.assert_close(out.detach(), h.detach(), **tol)
    h.backward(go)
    for a, p in zip(got, net.parameters()):
        gt = dict(rtol=1e-4, atol=1e-5) if not aggregate_first else dict(rtol=1e-3, atol=1e-4 * float(p.grad.abs().max()))
        torch.testing.assert_close(a, p.grad, **gt)


def _bench_worker(rank, world, port, tmp):
    import json
    import types

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gammagl_amd import benchmarks

        benchmarks.WORKLOADS["t400"] = dict(kind="gcn", dataset="t400", gen="rmat", also=None)
        benchmarks.DATASETS["t400"] = (400, 6000, 12, 5)
        args = types.SimpleNamespace(seed=0, relabel="random", order="src", hidden=16, layers=3, warmup=1,
                                     steps=2, workload="t400")
        out, ctx = benchmarks.run_gcn(args, torch.device("cpu"), rank, world, eng=_emul_engine())
        pg = ctx["pg"]
        assert out["n_gpus"] == world and out["value"] > 0 and out["scaling"] == "strong"
        assert out["rccl_ranks"] == world and pg.e_global == 6400
        # the headline is the reference's association: 2 aggregations per layer; the cheaper one is the side figure
        assert out["config"]["aggregations_per_step"] == 6 and out["config"]["aggregate_first"]["aggregations_per_step"] == 5
        # what the step puts on the wire, per step: layers 2 and 3 only (the input features' halo rows travel once)
        ex = out["config"]["exchange"]
        assert ex["a2a_calls"] == 4 and ex["halo_exposed_ms"] >= 0 and len(ex["a2a_isolated"]) == 2
        assert abs(ex["a2a_GB_in"] * 1e9 - (pg.n_halo + pg.n_send) * 4 * (16 + 8)) < 1e-3
        assert out["config"]["halo_floats_per_row_per_step"] == 2 * (16 + 8)
        rf = benchmarks.roofline_block("k", 4, 2.0, 10e9, 1e9, 1000, traffic_per_launch=1.5e9, traffic_source="t")
        assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "eff_GBps", "compulsory_bytes"}
        assert rf["frac"] <= 1.0 and abs(rf["achieved"] - 3000.0) < 1e-6 and rf["traffic_over_compulsory"] == 6.0
        json.dumps(out)
        # --relabel cluster at world > 1: the order is computed over the ranks' shares, the same bench body runs on it
        args.relabel, args.no_comparison, args.no_exchange_report = "cluster", True, True
        out2, ctx2 = benchmarks.run_gcn(args, torch.device("cpu"), rank, world, eng=_emul_engine())
        assert ctx2["pg"].e_global == 6400 and out2["value"] > 0 and "relabel=cluster" in out2["config"]["workload"]
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_bench_body_runs_distributed(tmp_path):
    """bench.py's body (per-rank graph construction, partition, timed steps, max-over-ranks) on gloo."""
    _run_ranks(2, tmp_path, "bench")


def _sage_worker(rank, world, port, tmp):
    """BASELINE config 4's multi-GPU mode: replicas that sample their own seeds + one gradient all-reduce."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gammagl_amd
        from gammagl_amd import torch_ops

        eng = _emul_engine()
        gammagl_amd._engine = eng                      # test-only: the layers' mpops calls land on the
        torch_ops.register_backend(lambda: eng, "CPU")  # host-emulated kernels in this subprocess
        from gammagl_amd.sampler import NeighborSampler
        from gammagl_amd.trainer import SAGETrainer

        N, F, Hd, C, ei, x, y, _ = _problem(2)
        ns = NeighborSampler(ei, [-1, -1], num_nodes=N, eng=eng)   # full neighbourhoods: deterministic blocks
        seeds_all = torch.randperm(N, generator=torch.Generator().manual_seed(11))[: 16 * world]
        mine = seeds_all[rank * 16:(rank + 1) * 16]
        tr = SAGETrainer(ns, F, Hd, C, num_layers=2, seed=5, device="cpu", group=None, world=world)
        l0 = tr.step(x, y, mine)
        # the same step in one process on the union of the seeds: mean of equal-sized means == mean
        ref = SAGETrainer(ns, F, Hd, C, num_layers=2, seed=5, device="cpu", world=1)
        l_ref = ref.step(x, y, seeds_all)
        lsum = l0.clone().reshape(1)
        dist.all_reduce(lsum)
        torch.testing.assert_close(lsum[0] / world, l_ref, rtol=1e-5, atol=1e-6)
        for p, q in zip(tr.net.parameters(), ref.net.parameters()):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-4, atol=1e-6)
        # sampled fan-outs: every replica draws its own blocks, the weights stay identical everywhere
        ns2 = NeighborSampler(ei, [5, 3], num_nodes=N, eng=eng)
        eng.reseed(100 + rank)
        tr2 = SAGETrainer(ns2, F, Hd, C, num_layers=2, seed=6, device="cpu", world=world)
        for it in range(3):
            sd = torch.randperm(N, generator=torch.Generator().manual_seed(20 + 7 * rank + it))[:24]
            assert torch.isfinite(tr2.step(x, y, sd))
        flat = torch.cat([p.detach().reshape(-1) for p in tr2.net.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        for other in gathered:
            assert torch.equal(other, flat)
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_sage_minibatch_replicas_with_gradient_allreduce(tmp_path):
    _run_ranks(2, tmp_path, "sage")


def _edge_set(src, dst):
    return set(zip(src.tolist(), dst.tolist()))


def _shard_worker(rank, world, port, tmp):
    """Config 5's construction path: every rank builds ONLY its share of the graph (no global edge list
    anywhere), the shares are exactly the world-1 graph cut at the same bounds, the halo bookkeeping built from
    them matches the one built by filtering the global list, a dry partition on one process reproduces this
    rank's buffers, and the fused halo epilogue equals aggregate -> bias_act bit for bit."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _emul_engine()
        from gammagl_amd.dist import PartitionedGraph, build_partition
        from gammagl_amd.synth import rmat_partitioned

        N, E_dir = 2000, 30000
        for relabel in ("random", "degree"):
            st = {}
            g = rmat_partitioned(N, E_dir, seed=5, rank=rank, world=world, relabel=relabel, stats=st)
            E = g["e_global"]
            assert E == E_dir + N
            # no rank ever holds the graph: the edge keys alive at any point of the build (its slice of the 1.5x
            # oversampled candidate stream, both directions, sent + received, before de-duplication) stay within
            # ~3x its final share, i.e. a 1/world fraction of the whole
            assert st["peak_edges"] <= 3.3 * E / world, (st, E, world)
            assert abs(st["local_edges"] - E / world) <= 0.25 * E / world + 800, st
            # the same graph for every world size: this share == the world-1 graph cut at the same bounds
            g1 = rmat_partitioned(N, E_dir, seed=5, rank=0, world=1, relabel=relabel)
            assert g1["e_global"] == E
            full_src, full_dst = g1["src"], g1["dst"]
            lo, hi = g["bounds"][rank], g["bounds"][rank + 1]
            from gammagl_amd.synth import bounds_from_degree
            assert g["bounds"] == bounds_from_degree(g1["deg"].long(), world)
            mine = (full_dst >= lo) & (full_dst < hi)
            assert _edge_set(g["src"], g["dst"] + lo) == _edge_set(full_src[mine], full_dst[mine])
            assert g["src"].numel() == int(mine.sum())          # no duplicates either
            torch.testing.assert_close(g["deg"], g1["deg"])
            # weights: symmetric GCN norm on the looped graph
            dis = g1["deg"].pow(-0.5)
            torch.testing.assert_close(g["w"], dis[g["src"]] * dis[g["dst"] + lo])
        # PartitionedGraph from the local share == PartitionedGraph from the filtered global list
        g = rmat_partitioned(N, E_dir, seed=5, rank=rank, world=world)
        g1 = rmat_partitioned(N, E_dir, seed=5)
        pg = PartitionedGraph.from_local(g["src"], g["dst"], g["w"], g["bounds"], N, g["e_global"], rank, world, eng=eng)
        ei1 = torch.stack([g1["src"], g1["dst"]])
        pgg = PartitionedGraph(ei1, g1["w"], N, rank, world, eng=eng, bounds=g["bounds"])
        assert pg.n_halo == pgg.n_halo and torch.equal(pg.halo_ids, pgg.halo_ids)
        assert torch.equal(pg.send_idx, pgg.send_idx) and pg.send_splits == pgg.send_splits
        assert pg.e_local == pgg.e_local and pg.e_global == pgg.e_global
        # a dry partition (one process playing this rank of `world`) has the same buffers and send lists
        pgd = build_partition(N, E_dir, 5, rank, 1, None, torch.device("cpu"), eng, parts=world)
        assert pgd.dry and pgd.n_halo == pg.n_halo and torch.equal(pgd.halo_ids, pg.halo_ids)
        assert pgd.send_splits == pg.send_splits and torch.equal(pgd.send_idx, pg.send_idx)
        assert pgd.lo == pg.lo and pgd.hi == pg.hi and pgd.e_local == pg.e_local
        hd = torch.randn(pgd.n_local, 8)
        assert pgd.aggregate(hd).shape == (pgd.n_local, 8)      # runs without a process group behind it
        # aggregate vs the unpartitioned op, and the fused halo epilogue vs aggregate -> bias_act
        for K in (16, 256):
            h = torch.randn(N, K, generator=torch.Generator().manual_seed(K))
            go = torch.randn(N, K, generator=torch.Generator().manual_seed(K + 1))
            bias = torch.randn(1, K, generator=torch.Generator().manual_seed(K + 2))
            hf = h.clone().requires_grad_(True)
            full = torch.relu(eng.c_spmm_sum(ei1, g1["w"], hf) + bias)
            full.backward(go)
            a = h[pg.lo:pg.hi].clone().requires_grad_(True)
            b1 = bias.clone().requires_grad_(True)
            ya = pg.aggregate(a, b1, relu=True)
            ya.backward(go[pg.lo:pg.hi])
            torch.testing.assert_close(ya.detach(), full.detach()[pg.lo:pg.hi], rtol=1e-5, atol=1e-4)
            torch.testing.assert_close(a.grad, hf.grad[pg.lo:pg.hi], rtol=1e-5, atol=1e-4)
            # dropout on: same rng state -> the fused form draws the mask of the two-pass form, bit for bit
            for relu in (True, False):
                eng.reseed(77)
                st0 = eng._rng_state(a.device).clone()
                a1 = h[pg.lo:pg.hi].clone().requires_grad_(True)
                b2 = bias.clone().requires_grad_(True)
                y1 = pg.aggregate(a1, b2, relu=relu, p_drop=0.4)
                y1.backward(go[pg.lo:pg.hi])
                eng._rng_state(a.device).copy_(st0)
                a2 = h[pg.lo:pg.hi].clone().requires_grad_(True)
                b3 = bias.clone().requires_grad_(True)
                y2 = eng.bias_act(pg.aggregate(a2), b3, relu=relu, p_drop=0.4)
                y2.backward(go[pg.lo:pg.hi])
                assert torch.equal(y1, y2) and 0.3 < float((y1 == 0).float().mean()) < (0.8 if relu else 0.5)
                assert torch.equal(a1.grad, a2.grad) and torch.equal(b2.grad, b3.grad)
                assert torch.equal(eng._rng_state(a.device), st0 + torch.tensor([0, 1]))
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_per_rank_graph_construction_and_fused_halo_epilogue(world, tmp_path):
    _run_ranks(world, tmp_path, "shard", timeout=600)


def test_rank_share_checks_at_toy_size():
    """The body of the -m gpu config-5 test (tests/share_checks.py) on a toy graph with the host-emulated kernels:
    rank 3 of an 8-way dry partition built piecewise (3 hash buckets), halo / send-list identities, the aggregate
    against f64 checksums and row evaluations, the adjoint identity through the reverse exchange."""
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    sys.path.insert(0, HERE)
    from share_checks import check_rank_share

    check_rank_share(_emul_engine(), torch.device("cpu"), 3000, 60000, P=8, r=3, min_buckets=3, buckets=3)
    check_rank_share(_emul_engine(), torch.device("cpu"), 2000, 30000, P=4, r=0)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts 2 ranks itself and reports n_gpus = 2
    (gloo + host-emulated kernels behind GGL_BENCH_EMUL); under a launcher with a different world size it
    refuses to print a line."""
    import json

    env = dict(os.environ, GGL_BENCH_EMUL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "tiny",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4000, r.stdout     # the headline alone, compact (bench.compact_line)
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["engine"].startswith("host-emulation")
    assert out["config"]["rank0_peak_edges_during_build"] <= 3.3 * 420000 / 2   # ~3x its share, transiently
    # one rank, unchanged contract
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--workload", "tiny",
                         "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    o1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert o1["n_gpus"] == 1 and o1["rccl_ranks"] == 1 and o1["config"]["parallelism"] == "1 GPU"
    # a launcher that started a different number of ranks than --gpus: no line
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--workload", "tiny"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                         timeout=120)
    assert bad.returncode != 0 and "{" not in bad.stdout
    # more ranks requested than this node has GPUs (none here): refuses instead of reporting an N-GPU line
    env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GGL_BENCH_EMUL")}
    few = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--workload", "tiny"], env=env2,
                         capture_output=True, text=True, timeout=120)
    if not torch.cuda.is_available() or torch.cuda.device_count() < 8:
        assert few.returncode != 0 and "{" not in few.stdout and "GPU" in (few.stderr + few.stdout)


def _cluster_worker(rank, world, port, tmp):
    """The locality order WITHOUT a global edge list: label sweeps over the ranks' shares (halo labels by all-to-all)
    give exactly the single-process `cluster_order`, `repartition` gives exactly the share `cut_share` cuts out of the
    renamed full edge list, and `build_partition(relabel="cluster")` at world > 1 goes that way."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _emul_engine()
        from gammagl_amd.dist import build_partition
        from gammagl_amd.partition import cluster_order, cluster_order_distributed
        from gammagl_amd.synth import _Comm, cut_share, planted_pairs, repartition, rmat_partitioned

        comm = _Comm(rank, world, None)
        # (a) a planted graph in a random labelling, cut into shares the way the R-MAT builder hands them out
        N = 3000
        src, dst = planted_pairs(N, out_deg=8, levels=((60, 0.7), (6, 0.2)), seed=2)
        pi = torch.randperm(N, generator=torch.Generator().manual_seed(9))
        src, dst = pi[src], pi[dst]
        g = cut_share(src, dst, N, parts=world, rank=rank)
        for C, arrange in ((24, True), (1500, False)):
            new_id, lab = cluster_order_distributed(g, comm, clusters=C, sweeps=8, seed=4, arrange=arrange)
            rk, lab1 = cluster_order(torch.stack([src, dst]), N, clusters=C, sweeps=8, seed=4, eng=eng, arrange=arrange,
                                     method="sort")
            lo, hi = g["bounds"][rank], g["bounds"][rank + 1]
            assert torch.equal(lab, lab1[lo:hi]), (C, int((lab != lab1[lo:hi]).sum()))
            assert torch.equal(new_id, rk[lo:hi])
            if C == 24:   # (and the dense-score form of the sweep agrees too: what the one-process build uses up to 1024 labels)
                rk2, _ = cluster_order(torch.stack([src, dst]), N, clusters=C, sweeps=8, seed=4, eng=eng, method="spmm")
                assert torch.equal(rk2, rk)
                g2 = repartition(g, new_id, comm)
                ref = cut_share(rk[src], rk[dst], N, parts=world, rank=rank)
                assert g2["bounds"] == ref["bounds"] and g2["e_global"] == ref["e_global"]
                for k in ("src", "dst", "w", "deg"):
                    assert torch.equal(g2[k], ref[k]), k
                # the order did its job: fewer remote sources than in the random labelling
                def remote(gg):
                    lo_, hi_ = gg["bounds"][rank], gg["bounds"][rank + 1]
                    t = torch.tensor([int(((gg["src"] < lo_) | (gg["src"] >= hi_)).sum())])
                    return int(comm.all_reduce(t))
                assert remote(g2) < 0.6 * remote(g), (remote(g2), remote(g))
        # (b) the product entry point on the R-MAT shares: same graph as the one-process build, rank's share of it
        st = {}
        pg = build_partition(2000, 30000, 5, rank, world, None, "cpu", eng, relabel="cluster", stats=st)
        g1 = rmat_partitioned(2000, 30000, seed=5, rank=0, world=1, relabel="random")
        s1, d1 = g1["src"][:-2000], g1["dst"][:-2000]
        rk, _ = cluster_order(torch.stack([s1, d1]), 2000, clusters=st["clusters"], sweeps=30, seed=5, eng=eng, method="sort")
        ref = cut_share(rk[s1], rk[d1], 2000, parts=world, rank=rank)
        assert pg.bounds == ref["bounds"] and pg.e_global == ref["e_global"]
        assert pg.n_local == ref["bounds"][rank + 1] - ref["bounds"][rank]
        x = torch.randn(2000, 8, generator=torch.Generator().manual_seed(1))
        lo, hi = pg.bounds[rank], pg.bounds[rank + 1]
        got = pg.aggregate(x[lo:hi].contiguous())
        exp = torch.zeros(hi - lo, 8).index_add_(0, ref["dst"], x[ref["src"]] * ref["w"].unsqueeze(1))
        assert torch.allclose(got, exp, rtol=1e-5, atol=1e-5)
        dist.barrier()
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_locality_order_over_rank_shares_matches_one_process(world, tmp_path):
    _run_ranks(world, tmp_path, "cluster")


if __name__ == "__main__":
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    mode = sys.argv[5] if len(sys.argv) > 5 else ""
    fn = {"bench": _bench_worker, "sage": _sage_worker, "shard": _shard_worker, "cluster": _cluster_worker}.get(mode, _worker)
    fn(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
