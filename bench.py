#!/usr/bin/env python3
"""bench.py — edges aggregated / second for a 3-layer GCN (hidden 256) training step on an
ogbn-products-sized synthetic graph, MI355X, through the gammagl_amd HIP kernels.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
With N > 1 and no launcher environment (WORLD_SIZE unset) bench.py launches the N ranks ITSELF
(torch.distributed.run, 127.0.0.1, one rank per GPU) and relays rank 0's line; under a launcher it checks
that WORLD_SIZE == --gpus and refuses to print a line for a different world size.  Every rank builds only
its own share of the graph (gammagl_amd.synth.rmat_partitioned): no rank holds the edge list.

* a "step" = one full training step of GCNModel(100 -> 256 -> 256 -> 47, norm='none') on
  precomputed symmetric-normalised edge weights (edge_weight = calc_gcn_norm(edge_index), the
  configuration examples/gcn/gcn_trainer.py:59 sketches) over the whole graph: forward (Linear,
  aggregate, +bias, ReLU, dropout per layer), softmax cross-entropy on the train nodes, backward, Adam
  with weight decay.  Every layer computes A (X W) as GammaGL's GCNConv does (gcn_conv.py:79): 3 forward
  CSR SpMMs + 3 transposed ones per step — that step is the line's `value` / `ms_per_step`.  The cheaper
  association (a layer whose input is narrower than its output computes (A X) W: 5 aggregations) is timed in
  the same run and reported beside it as `config.aggregate_first` (--aggregate-first makes it the main line).
  Same code path for every N (N = 1: no halo exchange);
* value = (aggregations per step) * E * steps / time over the whole job (max over ranks), inputs resident
  in HBM; scaling = "strong": the graph is fixed and node-partitioned over the N GPUs;
* both node orders of SURVEY.md §8d in ONE run: the random relabel (worst-case locality; the headline) and the
  degree-sorted one (`config.orderings`);
* roofline = the dominant kernel (CSR SpMM-sum, feature width 256, rank 0's largest edge block) timed with
  hipEvents on its launch stream.  `achieved` / `frac` = HBM-side bytes per launch MEASURED in this run
  (two rocprofv3 --pmc passes over `bench.py --pmc-probe`, same graph) / launch duration, against the 8 TB/s
  HBM3E peak: a fraction that cannot exceed 1.  The algorithmic rate E*(4K+8) + N*(4K+8) bytes per aggregate
  (no-reuse model) / duration is `eff_GBps` — it exceeds the peak when L2 serves gathers;
  `compulsory_bytes` (every row once) and `traffic_over_compulsory` say how much re-reading is left;
* cpu_baseline (N = 1 only) = the reference's own CPU extension (oracle/_ref, compiled from the
  reference sources; our C restatement if it is absent), 1 core (the shipped extension is serial:
  setup.py:50 never defines its OpenMP macro): ONE full-size K=256 aggregate of the benchmark graph itself
  (~35 s); `torch_fallback` = the reference's pure-torch formulation (mpops/torch.py:16-18,335-342) on an
  edge sample of the SAME graph, all host threads, median of 3;
* parity (N = 1): the HIP aggregate on the SAME graph, weights and features the cpu_baseline leg just ran the reference on,
  compared by oracle/parity.py (`rows_bit_exact_frac`, `max_rel_err`); the run exits 3 when it is outside the criterion;
* secondary (default command only): one full line (value, ms_per_step, roofline, cpu_baseline, parity) per remaining
  BASELINE config — arxiv, reddit-gat, sage-minibatch, papers-share — from child processes with short step counts;
* roofline fields are recomputable from the line: `traffic` = fabric-side bytes (FETCH_SIZE / WRITE_SIZE, Infinity-Cache
  hits included) of the row walk + the hub walk beside it, corrections calibrated IN THIS RUN on launches of known byte
  counts (`pmc_calibration`); `frac_of_peak`, `frac_of_achievable` (against the streaming read this part reaches, timed
  here), `alg_frac` (SURVEY §8d's no-reuse bytes / time / peak: may exceed 1), `frac` = min(frac_of_peak, 1);
* N > 1: the first collectives run on their own and a failing rank prints ONE diagnostic JSON line (stage, error, env);
  GGL_HALO_A2A=p2p switches the halo exchange to grouped isend / irecv;
* other workloads (--workload): arxiv | tiny | products-planted (a graph with community structure, random ids vs
  partition.cluster_order) | papers-share (config 5: one rank's share of the 8-way papers100M-sized partition on
  one GPU) | reddit-gat (config 3) | sage-minibatch (config 4) — same JSON shape;
* GGL_BENCH_EMUL=1 (tests only): gloo + the host-emulated kernels on CPU, to exercise the launcher and the
  N-rank code path where there is no GPU; the line then says "engine": "host-emulation" and is no measurement.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default=os.environ.get("GGL_BENCH_WORKLOAD", "products"),
                   help="products | arxiv | tiny | products-planted | papers-share | reddit-gat | sage-minibatch")
    p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--layers", type=int, default=3)
    p.add_argument("--order", default="src", choices=["src", "dst"],
                   help="edge order of the synthetic edge_index (src = coalesced COO as in GammaGL/PyG)")
    p.add_argument("--relabel", default="random", choices=["random", "degree", "none", "cluster"],
                   help="node order of the main line (cluster = partition.cluster_order on the randomly labelled graph)")
    p.add_argument("--also-relabel", default="auto", choices=["auto", "none", "random", "degree", "cluster"],
                   help="a second node order timed in the same run and reported in config.orderings (auto: degree for the "
                        "R-MAT workloads, cluster for products-planted; N = 1 only)")
    p.add_argument("--aggregate-first", action="store_true",
                   help="main line = the step in which a layer whose input is narrower than its output computes (A X) W "
                        "(default: A (X W) in every layer, as GammaGL's GCNConv writes it; the other association is the side figure)")
    p.add_argument("--transform-first", action="store_true", help="(accepted for compatibility: it is the default)")
    p.add_argument("--matmul-precision", default="highest", choices=["highest", "high"],
                   help="torch.set_float32_matmul_precision for the dense X W products: 'highest' = IEEE f32 MFMA (the "
                        "line of record); 'high' lets hipBLASLt emulate f32 with bf16 triples on gfx950 (2x faster GEMMs, "
                        "~5e-6 relative error instead of ~8e-7) — reported only as a side figure")
    p.add_argument("--no-tuned-gemm", action="store_true",
                   help="do not load gammagl_amd/tuned/*.csv (PyTorch TunableOp results: which rocBLAS / hipBLASLt f32 "
                        "kernel runs each GEMM shape of the step, chosen offline by tools/tune_gemms.sh)")
    p.add_argument("--hipgraph", default="auto", choices=["auto", "on", "off"],
                   help="GCN workloads on ONE rank: record the whole training step (fwd + bwd + Adam, dropout draws and "
                        "side-stream GEMMs included) into a hipGraph once and time its replays.  auto = graphs below 2^25 "
                        "edges, whose ~100 launches per step are issued slower than they execute; larger steps are "
                        "kernel-bound (zero gaps in the timeline) and run eagerly.  N > 1 steps are always eager (RCCL "
                        "collectives cannot be recorded on this stack, DESIGN.md §7)")
    p.add_argument("--no-comparison", action="store_true",
                   help="skip the other association and the second node order (profiling runs)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--pmc-traffic", default="auto", choices=["auto", "off", "l2"],
                   help="auto: after the timed region (N = 1), measure the dominant kernel's HBM-side traffic with two "
                        "rocprofv3 --pmc passes of `bench.py --pmc-probe` on the same graph (skipped when rocprofv3 is not "
                        "on PATH; falls back to the committed profile); l2: a third pass for the L2 hit rate")
    p.add_argument("--dry-parts", type=int, default=0,
                   help="N = 1 only: play ONE rank's share of a --dry-parts-way partition of the workload's graph as a dry "
                        "partition (send lists and buffers as in the real run, nothing on the wire): what a rank computes per "
                        "step, for tools/scaling_model.sh (papers-share is this with 8 parts built in)")
    p.add_argument("--dry-rank", type=int, default=-1, help="the rank played with --dry-parts (default: parts / 2)")
    p.add_argument("--secondary", default="auto", choices=["auto", "on", "off"],
                   help="auto: the default run (products, N = 1) also runs the other BASELINE configs — arxiv, reddit-gat, "
                        "sage-minibatch, papers-share — as child processes with short step counts and appends their full "
                        "lines as `secondary` (driver-timed figures for every config)")
    p.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# cpu_baseline legs: the ONLY place outside tests/ and smoke() that touches oracle/ — as the thing timed beside the
# GPU number, never as part of the product path
# ---------------------------------------------------------------------------------------------------------------
def _ref_or_port():
    from oracle import oracle as orc

    try:
        return orc.load_ref_ext(), "reference", orc
    except Exception:  # noqa: BLE001
        orc.build()
        return None, "port", orc


def _torch_fallback(ei, w, x, budget_s=10.0, chunk_edges=4_000_000):
    """The reference's pure-torch formulation of the same aggregate (mpops/torch.py:16-18,335-342: messages = x[src] * w ->
    zeros(N, K).scatter_add_(0, dst, messages)) on the benchmark graph itself, all host cores.  The [E, K] message tensor
    of the full graph is 129 GB at K = 256, so the edge list is walked in chunks of `chunk_edges` (the same three torch ops
    per chunk, ONE zero-filled output shared by all of them — round 4 zero-filled a fresh 2.5 GB output per 2 M-edge
    sample and reported that).  Thread count: the best of {8, 16, 32, 64, 128, all} on two chunks each (on the 256-thread
    host of the GPU box FEWER threads win: 32 -> 2.8 M edges/s, 256 -> 1.2 M; the random 1 KiB read-modify-writes into a
    2.5 GB output are latency-bound and the threads contend), then chunks until the time budget is used; value = edges done
    / (their time + the zero fill's share for that many edges).  It stays BELOW the reference's serial C++ loop on one core
    (3.6 M edges/s): that loop streams each message row once, torch materialises [chunk, K] messages and scatters them."""
    cores = os.cpu_count() or 1
    E, n = int(ei.shape[1]), int(x.shape[0])
    src_all, dst_all = ei[0], ei[1]

    def run_chunk(out, lo):
        hi = min(E, lo + chunk_edges)
        msg = x[src_all[lo:hi]] * w[lo:hi].view(-1, 1)
        out.scatter_add_(0, dst_all[lo:hi].view(-1, 1).expand_as(msg), msg)
        return hi - lo

    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    out = torch.zeros_like(x)
    t_zero = time.perf_counter() - t0
    run_chunk(out, 0)                                           # warm-up (allocator, thread pool)
    cands = sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores} or {cores})
    sweep = {}
    lo = chunk_edges
    for t in cands:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        done = 0
        for _ in range(2):
            if lo >= E:
                lo = 0
            done += run_chunk(out, lo)
            lo += chunk_edges
        sweep[t] = done / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t_run, e_done = 0.0, 0
    while t_run < budget_s and e_done < E:
        if lo >= E:
            lo = 0
        t0 = time.perf_counter()
        e_done += run_chunk(out, lo)
        t_run += time.perf_counter() - t0
        lo += chunk_edges
    torch.set_num_threads(cores)
    dt = t_run + t_zero * e_done / max(E, 1)
    return {"value": e_done / dt, "unit": "edges/s", "cores": best,
            "threads_swept": {str(k): round(v) for k, v in sweep.items()},
            "sample": f"pure-torch mpops formulation (x[src] * w -> scatter_add_), K={x.shape[1]}, the benchmark graph's own edge "
                      f"list in {chunk_edges}-edge chunks into ONE zero-filled [{n}, {x.shape[1]}] output: {e_done} of {E} edges in "
                      f"{t_run:.1f} s on {best} of {cores} threads (best of {cands}), zero fill {t_zero:.2f} s pro rata"}


def hip_parity_spmm(ei, w, x, y_ref, impl):
    """The `parity` object of a gcn line: the HIP CSR SpMM-sum (the step's dominant kernel, launched the way the step
    launches it: second sight of the weights = streamed from their sorted copy) on the tensors the CPU leg just ran the
    reference on, compared by oracle/parity.py (rows reduced in one piece bit-identical; chunk-combined hub rows within
    1e-5 of the row's magnitude)."""
    from gammagl_amd import engine
    from oracle import parity

    dev = torch.device("cuda", torch.cuda.current_device())
    eng = engine()
    ei_d, w_d, x_d = ei.to(dev), w.to(dev), x.to(dev)
    n = int(x.shape[0])
    gp = eng.graph_plan(ei_d, n)
    eng.c_spmm_sum(ei_d, w_d, x_d)
    y = eng.c_spmm_sum(ei_d, w_d, x_d)
    one = gp.fwd.counts() <= gp.fwd.chunk
    rep = parity.report(y, y_ref, rows_in_one_piece=one)
    # hub rows are added in the reference's serial order (hubf32.hip) unless the plan's longest row exceeds the exact walk's
    # bound: then EVERY row must be the reference's bits, not merely within the tolerance
    exact = int(eng.lib.ggl_get_option(b"exact_long_rows")) != 0 and int(gp.fwd.max_len) <= int(eng.lib.ggl_get_option(b"exact_long_max"))
    rep["exact_long_rows"] = exact
    if exact and rep["rows_bit_exact_frac"] < 1.0:
        rep["ok"] = False
        rep["why"] = "exact_long_rows is on: every f32 sum row must be bit-identical to the reference"
    rep["criterion"] = parity.CRITERION
    rep.update({"against": impl, "what": f"ONE K={int(x.shape[1])} CSR SpMM-sum forward of the benchmark graph itself "
                                         f"(N={n}, E={int(ei.shape[1])}), same weights and features on both sides",
                "rows_longer_than_chunk": int((~one).sum()), "chunk": int(gp.fwd.chunk),
                "col_block_launches": int(eng.lib.ggl_spmm_col_blocks(gp.E, int(x.shape[1]), n))})
    del ei_d, w_d, x_d, y, gp
    eng.clear_caches()
    torch.cuda.empty_cache()
    return rep


def cpu_baseline_gcn(hidden, classes, seed, full_graph=None):
    """Reference CPU extension (or the oracle port), 1 core.  `full_graph` = (edge_index [2,E] int64 on the
    host, weights [E], N): one K=hidden forward aggregate of the benchmark graph itself; the 6 aggregations of a step on
    a bounded R-MAT sample as a secondary figure; the pure-torch fallback on the same graph."""
    from gammagl_amd.synth import rmat_graph

    ref, kind, orc = _ref_or_port()
    impl = "reference c_spmm_sum (oracle/_ref)" if kind == "reference" else "oracle C port"
    cores = os.cpu_count() or 1
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(seed)
    out = {}
    x_full = None
    parity = None
    if full_graph is not None:
        ei, w, n = full_graph
        E = int(ei.shape[1])
        x_full = torch.randn(n, hidden, generator=gen)
        t0 = time.perf_counter()
        if ref is not None:
            y_ref = ref.c_spmm_sum(ei, w, x_full)
        else:
            y_ref = torch.from_numpy(orc.spmm_sum_fwd(ei.numpy(), w.numpy(), x_full.numpy()))
        dt = time.perf_counter() - t0
        # parity at the benchmark's own size: the HIP aggregate on the SAME graph, weights and features
        parity = hip_parity_spmm(ei, w, x_full, y_ref, impl)
        del y_ref
        out = {"value": E / dt, "unit": "edges/s", "cores": 1, "kind": kind,
               "sample": f"ONE forward aggregate (K={hidden}) of the full benchmark graph: N={n}, E={E}, {impl}, "
                         f"{dt:.1f} s on 1 core of {cores}"}
    n_s, e_s = 400000, (8_000_000 if full_graph is not None else 16_000_000)
    ei_s = rmat_graph(n_s, e_s, seed=seed + 17, device="cpu")
    E = ei_s.shape[1]
    w_s = torch.rand(E, generator=gen)
    widths = [hidden, hidden, classes]
    feats = [torch.randn(n_s, k, generator=gen) for k in widths]
    t0 = time.perf_counter()
    if ref is not None:
        eiT = ei_s.flip(0).contiguous()
        for x in feats:
            ref.c_spmm_sum(ei_s, w_s, x)        # forward aggregate (spmm_sum_cpu_forward)
            ref.c_spmm_sum(eiT, w_s, x)         # backward = the same loop on the transposed edge list
    else:
        ein, wn = ei_s.numpy(), w_s.numpy()
        for x in feats:
            orc.spmm_sum_fwd(ein, wn, x.numpy())
            orc.spmm_sum_bwd(ein, wn, x.numpy())
    dt = time.perf_counter() - t0
    step = {"value": 6 * E / dt, "unit": "edges/s", "cores": 1, "kind": kind,
            "sample": f"R-MAT N={n_s} E={E} (loops incl.), the 6 aggregations of one 3-layer GCN step "
                      f"(K={widths} fwd + transposed bwd), {impl}, {dt:.1f} s on 1 core of {cores}"}
    if out:
        out["step_sample"] = step
    else:
        out = step
    # second baseline: the reference's pure-torch formulation, same graph
    if full_graph is not None:
        out["torch_fallback"] = _torch_fallback(full_graph[0], full_graph[1], x_full)
    else:
        out["torch_fallback"] = _torch_fallback(ei_s, w_s, feats[0])
    return out, parity


def cpu_baseline_gat(ctx, seed):
    """Config 3's CPU counterpart: the reference ops composed as gat_conv.py:103-112 + softmax.py:29-35 write the
    layer (gather, LeakyReLU, c_segment_max, exp, c_segment_sum, divide, gather * alpha, c_segment_sum) for the 8 x 8
    head shape, on every 32nd edge of the benchmark graph (its own node set), 1 core."""
    ref, kind, orc = _ref_or_port()
    ei, n = ctx["ei"], ctx["n"]
    cores = os.cpu_count() or 1
    torch.set_num_threads(1)
    stride = 32
    ei = ei[:, ::stride].contiguous()
    H, C = 8, 8
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, H, C, generator=g)
    el, er = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    from oracle import parity as _parity

    # logits within 1e-4 of LeakyReLU's kink are left out of the comparison graph (oracle/parity.py kink_free_edges: an f32 logit on
    # the other side of 0 than its float64 value takes the other slope — a jump in the gradient that no precision removes)
    ei, n_kink = _parity.kink_free_edges_logits(ei, el, er)
    E = int(ei.shape[1])
    src, dst = ei[0], ei[1]
    go = torch.randn(n, H, C, generator=g)
    grads_ref = None
    t0 = time.perf_counter()
    if ref is not None:
        with torch.no_grad():      # the timed leg: the forward composition, as in rounds 3-4
            e = torch.nn.functional.leaky_relu(el[src] + er[dst], 0.2)
            m = ref.c_segment_max(e, dst, n)
            ex = torch.exp(e - m[dst])
            s = ref.c_segment_sum(ex, dst, n)
            alpha = ex / (s[dst] + 1e-16)
            y_ref = ref.c_segment_sum(x[src] * alpha.unsqueeze(-1), dst, n)
        impl = "reference c_segment_max / c_segment_sum (oracle/_ref) composed as gat_conv.py:103-112"
    else:
        y_ref = torch.from_numpy(orc.gat_fwd(ei.numpy(), el.numpy(), er.numpy(), x.numpy(), 0.2))
        impl = "oracle C port of the GATConv math"
    dt = time.perf_counter() - t0
    if ref is not None:            # untimed: the same composition under autograd, for the gradient comparison
        xb, elb, erb = (t.clone().requires_grad_(True) for t in (x, el, er))
        e = torch.nn.functional.leaky_relu(elb[src] + erb[dst], 0.2)
        m = ref.c_segment_max(e, dst, n)
        ex = torch.exp(e - m[dst])
        s = ref.c_segment_sum(ex, dst, n)
        alpha = ex / (s[dst] + 1e-16)
        ref.c_segment_sum(xb[src] * alpha.unsqueeze(-1), dst, n).backward(go)
        grads_ref = (xb.grad, elb.grad, erb.grad)
    from gammagl_amd import engine
    from oracle import parity as _par

    dev = torch.device("cuda", torch.cuda.current_device())
    ei_d = ei.to(dev)
    xa, ela, era = (t.to(dev).requires_grad_(True) for t in (x, el, er))
    y = engine().gat_fused(ei_d, ela, era, xa, 0.2)
    y.backward(go.to(dev))
    y = y.detach()
    par = _par.report(y, y_ref)
    par.update({"against": impl, "what": f"ONE fused GAT layer ({H} x {C}), forward + gradients, on every {stride}-th edge of the "
                                         f"benchmark graph ({E} edges, N={n}), same logits and features on both sides"})
    if grads_ref is not None:
        # gradients: HIP vs the reference's f32 composition (row-scale relative error), and BOTH against the layer in float64
        # (oracle/parity.py gat_truth_f64): criterion err(HIP) <= max(1e-5, 2 x err(reference f32 composition))
        hip = (y, xa.grad, ela.grad, era.grad)
        reff = (y_ref, *grads_ref)
        worst = 0.0
        for a, b, nm in zip(hip[1:], reff[1:], ("gx", "g_el", "g_er")):
            floor = float(b.abs().mean()) if nm != "gx" else 0.0
            worst = max(worst, _par.report(a, b, tol=1.0, floor_min=floor)["max_rel_err"])
        par["grad_max_rel_err"] = worst
        truth = _par.gat_truth_f64(ei_d, el.to(dev), er.to(dev), x.to(dev), go.to(dev), n)
        e_hip, e_ref = _par.gat_errors_vs_truth(truth, hip), _par.gat_errors_vs_truth(truth, reff)
        par["vs_fp64"] = {"hip": e_hip, "reference_f32": e_ref,
                          "criterion": "err(hip) <= max(1e-5, 2 x err(reference_f32)) for out, gx, g_el, g_er"}
        par["grad_tol"] = max(1e-5, 2.0 * max(e_ref[k] for k in ("gx", "g_el", "g_er")))
        par["grad_err_vs_fp64"] = max(e_hip[k] for k in ("gx", "g_el", "g_er"))
        par["ok"] = bool(par["ok"] and all(e_hip[k] <= max(1e-5, 2.0 * e_ref[k]) for k in e_hip))
        del truth
        # round 6: the MODEL of config 3 — 8 x 8 concat layer, ELU, head-averaging 41-class output layer (models/gat.py:36-72,
        # eval mode), i.e. the ggl_gat_sh_* kernels that carry 60 % of the step — forward + every parameter gradient on every
        # 64th edge against the reference ops composed (oracle/parity.py gat_model_composed) and against float64
        par["model"] = _gat_model_parity(ref, ctx["ei"], n, dev, seed)
        par["ok"] = bool(par["ok"] and par["model"]["ok"])
    par["criterion"] = _par.CRITERION
    engine().clear_caches()
    return {"value": E / dt, "unit": "edges/s", "cores": 1, "kind": kind,
            "sample": f"ONE GAT layer forward ({H} heads x {C} channels) over every {stride}-th edge of the benchmark graph "
                      f"({E} edges, N={n}), {impl}, {dt:.1f} s on 1 core of {cores}"}, par


def _gat_model_parity(ref, ei_host, n, dev, seed, stride=64):
    """GATModel(602, 8, 41, heads=8, 2 layers, fused) forward + parameter gradients on every `stride`-th edge of the benchmark
    graph: HIP vs the reference's c_segment_max / c_segment_sum composed as gat_conv.py:98-122 (f32, host) and vs the same
    composition in float64 (GPU).  Criterion per tensor: err(HIP) <= max(1e-5, 2 err(reference f32 composition))."""
    from gammagl_amd.layers import GATModel
    from oracle import parity as _par

    ei = ei_host[:, ::stride].contiguous()
    ei_d = ei.to(dev)
    torch.manual_seed(seed + 5)
    model = GATModel(602, 8, 41, heads=8, drop_rate=0.0, num_layers=2, fused=True).to(dev).eval()
    with torch.no_grad():
        for p in model.parameters():       # trained-like magnitudes (the default init's logits are ~0: a flat softmax)
            p.copy_(torch.randn_like(p) * (0.1 if p.dim() > 1 else 0.05))
    g = torch.Generator(device=dev).manual_seed(seed + 12)
    x = torch.randn(n, 602, generator=g, device=dev)
    go = torch.randn(n, 41, generator=g, device=dev)
    params = [(l.w, l.att, l.bias) for l in model.gat_list]
    with torch.no_grad():                  # near-kink edges of either layer left out (oracle/parity.py kink_free_edges_model)
        ei_d, n_kink = _par.kink_free_edges_model(ei_d, x, [tuple(p.detach() for p in tpl) for tpl in params], n, 8)
    ei = ei_d.cpu()
    y = model(x, ei_d, n)
    y.backward(go)
    hip = [y.detach()] + [p.grad for tpl in params for p in tpl]
    names = ["y"] + [f"g{nm}{li}" for li in range(2) for nm in ("W", "att", "b")]
    seg = ((lambda s_, ids, k: ref.c_segment_max(s_, ids, k)), (lambda v, ids, k: ref.c_segment_sum(v, ids, k)))

    def run(dtype, device, seg_ops):
        ps = [tuple(p.detach().to(device=device, dtype=dtype).requires_grad_(True) for p in tpl) for tpl in params]
        out = _par.gat_model_composed(x.to(device=device, dtype=dtype), ps, ei.to(device), n, 8, slope=0.2, seg=seg_ops)
        out.backward(go.to(device=device, dtype=dtype))
        return [out.detach()] + [p.grad for tpl in ps for p in tpl]

    torch.set_num_threads(1)
    reff = run(torch.float32, "cpu", seg)
    truth = run(torch.float64, dev, None)
    e_hip = _par.layer_errors_vs_truth(truth, hip, names)
    e_ref = _par.layer_errors_vs_truth(truth, reff, names)
    ok = all(e_hip[k] <= max(1e-5, 2.0 * e_ref[k]) for k in names)
    return {"ok": bool(ok), "what": f"GATModel 602 -> 8x8 (ELU) -> mean of 8x41, eval mode, forward + parameter gradients on every "
                                    f"{stride}-th edge ({int(ei.shape[1])} edges, N={n}; {n_kink} edges with a logit within 1e-4 of "
                                    f"LeakyReLU's kink left out)",
            "err_vs_fp64": {"hip": {k: float(f"{v:.3g}") for k, v in e_hip.items()},
                            "reference_f32": {k: float(f"{v:.3g}") for k, v in e_ref.items()}},
            "criterion": "err(hip) <= max(1e-5, 2 x err(reference_f32)) per tensor; errors are row-scale relative"}


def cpu_baseline_sage(ctx, hidden, seed):
    """Config 4's CPU counterpart: the reference's c_segment_mean on messages shaped like the batch's blocks
    ([edges, hidden] -> [destination rows, hidden]), 1 core, 20 repetitions."""
    ref, kind, orc = _ref_or_port()
    cores = os.cpu_count() or 1
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(seed)
    tot_e, reps = 0, 20
    t_all = 0.0
    from gammagl_amd import engine
    from oracle import parity as _par

    dev = torch.device("cuda", torch.cuda.current_device())
    par = None
    for blk, (n_src, n_e) in zip(ctx["blocks"], ctx["valid"]):
        n_dst = int(blk.n_dst_cap)
        dst = torch.sort(torch.randint(0, n_dst, (n_e,), generator=g)).values
        msg = torch.randn(n_e, hidden, generator=g)
        t0 = time.perf_counter()
        for _ in range(reps):
            if ref is not None:
                y_ref = ref.c_segment_mean(msg, dst, n_dst)
            else:
                y_ref = torch.from_numpy(orc.segment_mean(msg.numpy(), dst.numpy(), n_dst))
        t_all += time.perf_counter() - t0
        tot_e += n_e
        r = _par.report(engine().c_segment_mean(msg.to(dev), dst.to(dev), n_dst), y_ref)
        if par is None or r["max_rel_err"] > par["max_rel_err"] or not r["ok"]:
            par = r
    impl = "reference c_segment_mean (oracle/_ref)" if kind == "reference" else "oracle C port"
    return {"value": tot_e * reps / t_all, "unit": "edges/s", "cores": 1, "kind": kind,
            "sample": f"segment_mean of [edges, {hidden}] messages shaped like one batch's two sampled blocks ({tot_e} edges), "
                      f"{reps} repetitions, {impl}, {t_all:.1f} s on 1 core of {cores}"}, \
        dict(par, against=impl, criterion=_par.CRITERION, what=f"unsorted_segment_mean of [edges, {hidden}] messages shaped like the batch's two sampled "
                                     f"blocks (the worse of the two reported)")


# ---------------------------------------------------------------------------------------------------------------
# measured HBM traffic of the dominant kernel (rocprofv3 --pmc around `bench.py --pmc-probe`)
# ---------------------------------------------------------------------------------------------------------------
def measure_traffic(args, kernel_substrs, relabel=None, with_l2=False):
    """Fabric-side bytes per launch of the dominant kernel, measured NOW on this box: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section: KiB units) of `bench.py --pmc-probe`, which
    first launches two CALIBRATION patterns of known byte counts (benchmarks.calibration_launches: a 16 B/lane streaming
    copy and a gather of 256-byte rows through a random permutation), then rebuilds this run's graph and launches the
    kernel a few times.  The read-side correction applied to the kernel is the one MEASURED on the 256-byte gather (the
    aggregate's own access pattern), the write-side one the one measured on the streaming copy — not a constant.
    Returns (bytes, source, extra) or (None, reason, {})."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    from gammagl_amd.benchmarks import CALIB, CALIB_ORDER, calib_known_bytes

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH", {}
    vals, calib = {}, {}
    # the third pass is optional evidence: how many of the L2's fabric-side read requests are tagged "destined for DRAM
    # (MC)" as opposed to GMI / IO — NOT a post-Infinity-Cache count (the MALL sits behind the same port)
    dram = ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum")
    passes = [("FETCH_SIZE",), ("WRITE_SIZE",)] + ([("TCC_HIT_sum", "TCC_MISS_sum")] if with_l2 else []) + \
        ([dram] if os.environ.get("GGL_BENCH_DRAM_PASS", "1") == "1" else [])
    for counters in passes:
        d = tempfile.mkdtemp(prefix="ggl_pmc_")
        try:
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-probe", "--workload", args.workload, "--hidden", str(args.hidden),
                   "--seed", str(args.seed), "--relabel", relabel or args.relabel, "--order", args.order,
                   "--dry-parts", str(args.dry_parts), "--dry-rank", str(args.dry_rank)]
            r = subprocess.run(cmd, cwd=d, env=dict(os.environ, TMPDIR=d), capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if counters is dram:
                    continue
                return None, f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode}): {r.stderr[-200:]}", {}
            got, cal = counters_per_launch(files[0], r.stdout, kernel_substrs, counters)
            missing = [c for c, v in got.items() if v is None]
            if missing:
                if counters is dram:
                    continue
                return None, f"kernel '{kernel_substrs[0]}' not found in the {missing[0]} counter file", {}
            vals.update(got)
            reps = CALIB["reps"]
            for c, per in cal.items():     # the FIRST `reps` dispatches of each calibration kernel (the probe runs them first)
                got = {w: sorted(v for _, v in sorted(xs)[:reps])[reps // 2] for w, xs in per.items() if len(xs) >= reps}
                if len(got) == len(CALIB_ORDER):
                    calib[c] = got
        except Exception as ex:  # noqa: BLE001
            return None, f"{type(ex).__name__}: {ex}", {}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    extra = {}
    if with_l2:
        h, m = vals["TCC_HIT_sum"], vals["TCC_MISS_sum"]
        extra["l2_hit_rate"] = h / (h + m) if h + m > 0 else None
    # corrections: known bytes / (counter x 1 KiB) on the calibration launches of this very run
    rd_f, wr_f, how = 2.0, 1.0, "guide constants (x2 read side, x1 write side): calibration rows missing from the counter files"
    if "FETCH_SIZE" in calib and "WRITE_SIZE" in calib:
        cal_out = {}
        for which in CALIB_ORDER:
            rd, wr = calib_known_bytes(which)
            f_kib, w_kib = calib["FETCH_SIZE"][which], calib["WRITE_SIZE"][which]
            cal_out[which] = {"known_read_bytes": rd, "FETCH_SIZE_KiB": f_kib, "read_factor": rd / max(f_kib * 1024.0, 1.0),
                              "known_write_bytes": wr, "WRITE_SIZE_KiB": w_kib,
                              "write_factor": (wr / max(w_kib * 1024.0, 1.0)) if wr else None}
        rd_f, wr_f = cal_out["gather256"]["read_factor"], cal_out["stream_copy"]["write_factor"]
        cal_out["applied"] = {"read_factor": rd_f, "write_factor": wr_f,
                              "rule": "bytes = read_factor(gather256) x FETCH_SIZE x 1024 + write_factor(stream_copy) x WRITE_SIZE x 1024"}
        extra["pmc_calibration"] = cal_out
        how = "corrections measured in this run on launches of known byte counts (roofline.pmc_calibration)"
    extra["pmc_raw"] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"]}
    if all(c in vals for c in dram):
        extra["dram_destined_requests"] = {
            "TCC_EA0_RDREQ": vals[dram[0]], "TCC_EA0_RDREQ_DRAM": vals[dram[1]], "TCC_EA0_WRREQ_DRAM": vals[dram[2]],
            "read_share_destined_for_dram": vals[dram[1]] / max(vals[dram[0]], 1.0),
            "note": "requests the L2 sends towards local memory (MC) per launch; the Infinity Cache sits behind that port, "
                    "so this is a routing tag, not a DRAM-only byte count — gfx950 exposes no post-MALL counter to rocprofv3"}
    return (rd_f * vals["FETCH_SIZE"] + wr_f * vals["WRITE_SIZE"]) * 1024.0, \
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --pmc-probe` on the same graph; " + how, extra


def counters_per_launch(csv_path, probe_stdout, kernel_substrs, counters):
    """One rocprofv3 counter file of `bench.py --pmc-probe` -> ({counter: value per LAUNCH of the dominant walk}, {counter:
    {calibration launch: [values]}}).  The probe's own launches are the LAST dispatches of each kernel: its stdout says
    `aggregates=A dispatches=a,b` — a dispatches of kernel_substrs[0] (the row walk: once per column block), b of
    kernel_substrs[1] (the hub walk beside it: once per AGGREGATE since round 5).  A counter's total over one aggregate =
    sum over the kernels of (its last dispatches) / A; per launch = / (a / A), like ms_per_launch = ms_per_aggregate / launches."""
    import csv
    import re

    from gammagl_amd.benchmarks import CALIB, CALIB_ORDER

    m = re.search(r"dispatches=([\d,]+)", probe_stdout)
    lasts = [int(v) for v in m.group(1).split(",")] if m else []
    m = re.search(r"aggregates=(\d+)", probe_stdout)
    aggs = int(m.group(1)) if m else 0
    acc = {c: [[] for _ in kernel_substrs] for c in counters}
    cal = {c: {w: [] for w in CALIB_ORDER} for c in counters}
    with open(csv_path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] not in acc:
                continue
            item = (int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"]))
            hit = [i for i, k in enumerate(kernel_substrs) if k in row["Kernel_Name"]]
            if hit:
                acc[row["Counter_Name"]][hit[0]].append(item)
            else:
                for which in CALIB_ORDER:
                    if CALIB[which]["kernel"] in row["Kernel_Name"]:
                        cal[row["Counter_Name"]][which].append(item)
    vals = {}
    for c, per_kernel in acc.items():
        if not per_kernel[0]:
            vals[c] = None
            continue
        tot = 0.0
        for i, xs in enumerate(per_kernel):
            xs = [v for _, v in sorted(xs)]
            last = lasts[i] if i < len(lasts) else (lasts[0] if lasts else 0)
            xs = xs[-last:] if 0 < last <= len(xs) else xs
            tot += (sum(xs) / aggs) if aggs > 0 else ((sum(xs) / len(xs)) if xs else 0.0)
        per_agg = max(lasts[0] // aggs, 1) if (aggs > 0 and lasts) else 1
        vals[c] = tot / per_agg
    return vals, cal



def committed_traffic(args, launches, E):
    """Fallback when rocprofv3 cannot run here: the committed --pmc profile of THIS workload (never another one's)."""
    if not (args.workload == "products" and args.hidden == 256 and args.order == "src" and args.relabel == "random"
            and args.seed == 0):
        return None, None
    for name in ("r6_pmc_products_k256.json", "r2_pmc_products_k256.json"):
        try:
            rec = json.load(open(os.path.join(REPO, "profiles", name)))
            if int(rec.get("graph_edges", -1)) == E and int(rec.get("launches_per_aggregate", 1)) == launches:
                return rec["spmm_sum_k256"]["hbm_bytes_per_launch"], \
                    f"profiles/{name} (rocprofv3 --pmc passes on this graph, NOT collected in this run)"
        except Exception:  # noqa: BLE001
            continue
    return None, None


# kernel-name fragments of the dominant launch in a counter file: the row walk + (gcn) the launch that adds up the hub rows
# in serial order beside it (hubf32.hip: one dispatch of each per column-block launch; absent when the plan has no long rows)
KERNEL_OF = {"gcn": ("row_reduce_kernel<float, 4, 0, 1,", "hub_rows_f32_kernel<"), "gat": ("gat_fwd2_kernel",)}


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(args):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run on this node
    (one per GPU, rendezvous on 127.0.0.1) and relay their output.  Refuses when the node has fewer GPUs."""
    import subprocess

    emul = os.environ.get("GGL_BENCH_EMUL") == "1"
    if not emul:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); not reporting a "
                             f"{args.gpus}-GPU line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


# ---------------------------------------------------------------------------------------------------------------
# the other BASELINE configs, driver-timed: the default run appends one full line per config as `secondary`
# ---------------------------------------------------------------------------------------------------------------
SECONDARY = [   # (workload, BASELINE.json config, extra flags): short step counts, same code paths as --workload X
    ("arxiv", "configs[1]: 3-layer GCN hidden=256 on ogbn-arxiv", ["--steps", "30", "--warmup", "5"]),
    ("reddit-gat", "configs[2]: 8-head GAT on Reddit", ["--steps", "5", "--warmup", "2"]),
    ("sage-minibatch", "configs[3]: GraphSAGE neighbour-sampled mini-batches on ogbn-products", ["--steps", "60", "--warmup", "10"]),
    ("papers-share", "configs[4]: one rank's share of the 8-way papers100M-sized partition (dry)", ["--steps", "3", "--warmup", "1"]),
]
SECONDARY_BUDGET_S = float(os.environ.get("GGL_BENCH_BUDGET_S", "340"))   # whole-command wall clock aimed at (~6 min)


def secondary_wanted(args, world, emul):
    if args.secondary == "off" or world != 1 or emul:
        return False
    return args.secondary == "on" or (args.workload == "products" and not args.dry_parts)


def run_secondary(args, t_start):
    """One child `bench.py --workload X` per remaining BASELINE config (this process has released the GPU's memory): each
    returns its full line (value, ms_per_step, roofline, cpu_baseline, parity).  A child that fails or would overrun the
    command's time budget is reported as such, never silently dropped."""
    import subprocess
    import tempfile

    lines = []
    for name, config, flags in SECONDARY:
        spent = time.perf_counter() - t_start
        if spent > SECONDARY_BUDGET_S:
            lines.append({"workload": name, "baseline_config": config,
                          "skipped": f"time budget: {spent:.0f} s of {SECONDARY_BUDGET_S:.0f} s used before it started"})
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", name, "--secondary", "off",
               "--no-comparison", "--seed", str(args.seed), "--hidden", str(args.hidden)] + flags
        if name == "papers-share":
            cmd += ["--pmc-traffic", "off"]    # (two more builds of a 93 GB share: its counters live in profiles/)
        t0 = time.perf_counter()
        detail = os.path.join(tempfile.gettempdir(), f"ggl_bench_{os.getpid()}_{name}.json")
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=max(60.0, SECONDARY_BUDGET_S + 120 - spent),
                               env=dict(os.environ, GGL_BENCH_DETAIL=detail))
            js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            line = json.loads(js[-1]) if js else None
            if line is not None and os.path.exists(detail):     # the child's complete record (its stdout line is compact)
                with open(detail) as f:
                    line = json.load(f)
                os.unlink(detail)
            if line is None:
                line = {"workload": name, "error": f"no JSON line (rc {r.returncode}): {r.stderr[-300:]}"}
            elif r.returncode != 0:
                line["rc"] = r.returncode
        except subprocess.TimeoutExpired:
            line = {"workload": name, "error": "timed out"}
        except Exception as ex:  # noqa: BLE001
            line = {"workload": name, "error": f"{type(ex).__name__}: {ex}"}
        line["baseline_config"] = config
        line["wall_s"] = round(time.perf_counter() - t0, 1)
        lines.append(line)
    return lines


# ---------------------------------------------------------------------------------------------------------------
# the line the driver parses: compact.  Everything verbose (prose, calibration tables, the secondary configs' full
# lines) goes to bench_detail.json + earlier stdout lines; the LAST stdout line is the headline alone, < 4 KB
# ---------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4000      # bytes; the driver keeps a bounded tail of stdout (round 4's 28 KB line was not parseable)
DETAIL_FILE = os.environ.get("GGL_BENCH_DETAIL", os.path.join(REPO, "bench_detail.json"))


def _r(v, nd=4):
    """numbers at the precision they are measured to (shorter line, same content)"""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{nd + 2}g}") if abs(v) >= 1 else round(v, nd + 3)
    return v


def _pick(d, keys, nd=4):
    return {k: _r(d[k], nd) for k in keys if d is not None and k in d and d[k] is not None}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def compact_secondary(line):
    """what the headline keeps of a secondary config's full line"""
    if "error" in line or "skipped" in line:
        return _pick(line, ("workload", "error", "skipped")) | {"workload": line.get("workload")}
    rf, par = line.get("roofline") or {}, line.get("parity") or {}
    return {"workload": _short(line.get("config", {}).get("workload", line.get("workload", "?")).split(":")[0], 24),
            "value": _r(line.get("value")), "unit": "edges/s", "ms_per_step": _r(line.get("ms_per_step")),
            "frac": _r(rf.get("frac")), "alg_frac": _r(rf.get("alg_frac")), "parity_ok": par.get("ok"),
            **({"model_parity_ok": par["model"].get("ok")} if isinstance(par.get("model"), dict) else
               ({"model_parity_ok": par["model_ok"]} if "model_ok" in par else {})),
            "cpu_baseline": _r((line.get("cpu_baseline") or {}).get("value"))}


def compact_line(out):
    """The contract's ONE line: every field the driver and the judge read (metric, value, ms_per_step, steps, warmup,
    dtype, config, roofline, cpu_baseline, parity), numbers only — the prose is in bench_detail.json."""
    c = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data", "rccl_ranks", "engine") if k in out}
    cfg = out.get("config") or {}
    cc = {"workload": _short(cfg.get("workload", ""), 200)}
    cc.update(_pick(cfg, ("association", "hipgraph", "aggregations_per_step", "parallelism", "gcn_norm", "rank0_local_edges",
                          "rank0_owned_rows", "rank0_halo_rows", "rank0_send_rows", "rank0_peak_edges_during_build",
                          "rank0_halo_GB_per_step", "matmul_precision", "loss", "batch",
                          "launches_per_step")))
    if cfg.get("route"):
        cc["route"] = _short(str(cfg["route"]), 16)
    if isinstance(cfg.get("routes"), dict):
        cc["routes_ms"] = {("cpp" if "ggl" in k else "ctypes"): v for k, v in cfg["routes"].items() if k != "unit"}
    if "engine" in c:
        c["engine"] = _short(c["engine"], 40)
    if "hipgraph" in cc:
        cc["hipgraph"] = _short(cc["hipgraph"], 24)
    if "matmul_precision" in cc:
        cc["matmul_precision"] = _short(cc["matmul_precision"], 24)
    af = cfg.get("aggregate_first") or cfg.get("transform_first")
    if isinstance(af, dict):
        cc["aggregate_first" if "aggregate_first" in cfg else "transform_first"] = _pick(af, ("ms_per_step", "value", "aggregations_per_step"))
    if cfg.get("orderings"):
        cc["orderings"] = [_pick(o, ("relabel", "ms_per_step", "value", "ms_per_aggregate_K256")) for o in cfg["orderings"]]
    if isinstance(cfg.get("norm_both"), dict):
        cc["norm_both"] = _pick(cfg["norm_both"], ("ms_per_step_cached", "ms_per_step_uncached"))
    ex = cfg.get("exchange")
    if isinstance(ex, dict):
        cc["exchange"] = _pick(ex, ("halo_exposed_ms", "a2a_calls", "a2a_GB_in", "a2a_GB_out", "a2a_isolated_ms_per_step", "overlap_frac"))
        iso = ex.get("a2a_isolated") or []
        if iso:      # the widest layer's exchange on its own: GB/s per link (the figure the scaling hinges on)
            cc["exchange"]["widest"] = _pick(max(iso, key=lambda e: e.get("K", 0)), ("K", "chunks", "fwd_ms", "bwd_ms", "GBps_per_link_fwd"))
        if ex.get("halo_chunks"):
            cc["exchange"]["halo_chunks"] = ex["halo_chunks"] if not isinstance(ex["halo_chunks"], dict) else \
                {k: v for k, v in ex["halo_chunks"].items() if k in ("exchange", "const")}
    c["config"] = cc
    rf = out.get("roofline")
    if rf:
        r = {"bound": rf.get("bound", "hbm"), "kernel": _short(rf.get("kernel", ""), 100)}
        r.update(_pick(rf, ("achieved", "peak", "unit", "frac", "traffic", "launches_per_aggregate", "ms_per_launch", "ms_per_aggregate",
                            "alg_bytes_per_aggregate", "eff_GBps", "alg_frac", "frac_of_peak", "achievable_GBps", "frac_of_achievable",
                            "compulsory_bytes", "traffic_over_compulsory", "traffic_over_algorithmic", "hub_ms_per_launch",
                            "row_walk_ms_per_launch")))
        r["traffic_source"] = _short(rf.get("traffic_source", ""), 90)
        cal = (rf.get("pmc_calibration") or {}).get("applied")
        if cal:
            r["pmc_factors"] = _pick(cal, ("read_factor", "write_factor"))
        c["roofline"] = r
    cb = out.get("cpu_baseline")
    if cb:
        b = _pick(cb, ("value", "unit", "cores", "kind"))
        b["sample"] = _short(cb.get("sample", ""), 160)
        tf = cb.get("torch_fallback")
        if isinstance(tf, dict):
            b["torch_fallback"] = _pick(tf, ("value", "cores", "unit"))
            b["torch_fallback"]["sample"] = _short(tf.get("sample", ""), 120)
        c["cpu_baseline"] = b
    par = out.get("parity")
    if par:
        pp = _pick(par, ("ok", "rows", "rows_bit_exact_frac", "elems_bit_exact_frac", "tol", "max_rel_err", "max_abs_err",
                         "rows_longer_than_chunk", "grad_max_rel_err", "grad_err_vs_fp64", "grad_tol", "bwd_rows_bit_exact_frac"), nd=3)
        pp["against"] = _short(par.get("against", ""), 60)
        if par.get("criterion"):
            pp["criterion"] = "row-scale rel err <= tol; one-piece rows bit-identical"
        if isinstance(par.get("model"), dict):
            m = par["model"]
            pp["model_ok"] = m.get("ok")
            ev = m.get("err_vs_fp64") or {}
            if ev:
                pp["model_err_vs_fp64"] = {k: max(v.values()) for k, v in ev.items()}
        c["parity"] = pp
    if out.get("secondary"):
        c["secondary"] = [compact_secondary(x) for x in out["secondary"]]
    c["detail"] = os.path.basename(DETAIL_FILE)
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > LINE_LIMIT:      # never let prose push the contract's fields out of the driver's tail
        for key in ("traffic_source", "sample"):
            for obj in (c.get("roofline", {}), c.get("cpu_baseline", {}), c.get("cpu_baseline", {}).get("torch_fallback", {})):
                if key in obj:
                    obj[key] = _short(obj[key], 40)
        c["config"]["workload"] = _short(c["config"]["workload"], 60)
        line = json.dumps(c, separators=(",", ":"))
    # still too long (cannot happen with today's fields): drop optional side figures one by one — never the contract's
    # fields, and never raise: a line that is a little long is better than no line
    for sect, key in (("config", "exchange"), ("config", "orderings"), ("config", "norm_both"), ("config", "aggregate_first"),
                      (None, "secondary")):
        if len(line) <= LINE_LIMIT:
            break
        (c if sect is None else c.get(sect, {})).pop(key, None)
        line = json.dumps(c, separators=(",", ":"))
    return line


def emit(out):
    """stdout: one compact line per secondary config (prefixed by nothing: each is a valid JSON line), then the
    headline LAST; the complete, verbose record -> bench_detail.json (and gpurun_out/ when that directory exists)."""
    try:
        paths = {DETAIL_FILE} | ({os.path.join(REPO, "gpurun_out", "bench_detail.json")} if "GGL_BENCH_DETAIL" not in os.environ else set())
        for path in paths:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    json.dump(out, f, indent=1)
    except OSError as ex:
        print(f"bench.py: could not write the detail file: {ex}", file=sys.stderr)
    def safe(line_of):
        try:
            return compact_line(line_of)
        except Exception as ex:  # noqa: BLE001 — the contract's fields must reach stdout whatever an optional field holds
            core = {k: line_of.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
            core["config"] = {"workload": _short((line_of.get("config") or {}).get("workload", ""), 200)}
            core["compact_line_error"] = f"{type(ex).__name__}: {ex}"[:200]
            return json.dumps(core, separators=(",", ":"))

    for sec in out.get("secondary") or []:
        if "metric" in sec:
            print(safe(sec), flush=True)
    print(safe(out), flush=True)


def main():
    t_start = time.perf_counter()
    args = parse()
    from gammagl_amd.benchmarks import PROBES, RUNNERS, WORKLOADS, set_traffic, sizes_of

    if args.workload not in WORKLOADS:
        raise SystemExit(f"bench.py: unknown --workload {args.workload}; choose from {', '.join(WORKLOADS)}")
    kind = WORKLOADS[args.workload]["kind"]
    if args.pmc_probe:   # child of measure_traffic: the dominant kernel alone, under rocprofv3 --pmc
        from gammagl_amd import engine

        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        PROBES[kind](args, dev, engine())
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    emul = os.environ.get("GGL_BENCH_EMUL") == "1"
    eng = None
    if emul:  # tests only: the launcher + N-rank path on a GPU-less box (gloo, host-emulated kernels)
        import subprocess

        from gammagl_amd import _lib
        from gammagl_amd.ops import Engine

        subprocess.check_call([os.path.join(REPO, "tests", "emul", "build.sh")])
        eng = Engine(_lib.bind(os.path.join(REPO, "tests", "emul", "libggl_emul.so")), require_cuda=False)
        dev, backend = torch.device("cpu"), "gloo"
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
        torch.cuda.set_device(local_rank)
        dev, backend = torch.device("cuda", local_rank), "nccl"
    stage = {"name": "start"}

    def diagnostic(err):
        """N > 1 only: what failed and where, as ONE JSON line per failing rank (the first multi-GPU run of this code is
        the driver's: a bare traceback from rank 5 of 8 says little).  `stage` = the last step this rank had reached."""
        import traceback

        line = {"diagnostic": True, "metric": "bench.py did not produce a measurement", "stage": stage["name"], "rank": rank,
                "world": world, "backend": backend, "error": f"{type(err).__name__}: {err}"[:3000],
                "traceback_tail": traceback.format_exc()[-1500:],
                "halo_exchange": os.environ.get("GGL_HALO_A2A", "a2a"),
                "env": {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK", "NCCL_DEBUG",
                                                        "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME", "HIP_VISIBLE_DEVICES")},
                "hint": "GGL_HALO_A2A=p2p switches the halo exchange from all_to_all_single to grouped isend / irecv; "
                        "NCCL_DEBUG=INFO prints RCCL's own account to stderr"}
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("NCCL_DEBUG", "WARN")          # RCCL's error text on stderr, not just an error code
        try:
            stage["name"] = "init_process_group"
            if emul:
                dist.init_process_group(backend)
            else:
                dist.init_process_group(backend, device_id=dev)
            # the first collectives, on their own: an all-reduce, then the uneven all-to-all-v the halo exchange is made of
            stage["name"] = "first all_reduce"
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            assert int(probe.item()) == world, f"all_reduce of ones over {world} ranks returned {probe.item()}"
            stage["name"] = "first uneven all_to_all_single"
            send = torch.full((sum(range(1, world + 1)), 4), float(rank), device=dev)
            splits_in = list(range(1, world + 1))
            splits_out = [rank + 1] * world
            recv = torch.empty((sum(splits_out), 4), device=dev)
            dist.all_to_all_single(recv, send, splits_out, splits_in)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            want = torch.arange(world, device=dev, dtype=torch.float32).repeat_interleave(rank + 1)
            assert torch.equal(recv[:, 0], want), "uneven all_to_all_single delivered the wrong rows"
            stage["name"] = "benchmark body"
        except Exception as err:  # noqa: BLE001
            diagnostic(err)
            raise

    torch.set_float32_matmul_precision(args.matmul_precision)
    tuned = None
    if not emul and not args.no_tuned_gemm and args.matmul_precision == "highest":
        # library GEMM selection only: the same IEEE f32 products, each shape on the rocBLAS / hipBLASLt kernel an offline
        # TunableOp pass measured fastest on this part (the file's validators — torch / HIP / library versions, gfx
        # arch — must match, otherwise torch ignores it and the default heuristics pick)
        names = [args.workload.replace("-", "_")] + ([WORKLOADS[args.workload]["dataset"] or "tiny"] if kind == "gcn" else [])
        paths = [os.path.join(REPO, "gammagl_amd", "tuned", f"tunableop_gfx950_{n}.csv") for n in names]
        path = next((p for p in paths if os.path.exists(p)), None)   # the workload's own file, else its dataset's
        if path is not None:
            try:
                import torch.cuda.tunable as tun

                tun.enable(True)
                tun.tuning_enable(False)
                tuned = os.path.relpath(path, REPO) if tun.read_file(path) else None
                if tuned is None:
                    tun.enable(False)
            except Exception:  # noqa: BLE001
                tuned = None

    want_cpu = world == 1 and not args.no_cpu_baseline and not emul
    args.keep_host_graph = want_cpu and kind == "gcn"
    try:
        out, ctx = RUNNERS[kind](args, dev, rank, world, eng=eng)
    except Exception as err:  # noqa: BLE001
        if world > 1:
            diagnostic(err)
        raise
    route = (out.get("config") or {}).get("route")
    out["engine"] = ("host-emulation (launcher test, not a measurement)" if emul else
                     ("torch.ops.ggl (dispatcher -> libggl_torch.so -> C ABI -> libggl_mpops_hip.so: the route compat/_torch_ext.py binds)"
                      if route == "cpp" else "hip (ctypes engine -> C ABI -> libggl_mpops_hip.so)"))
    out["config"]["tuned_gemm_selection"] = tuned
    out["config"]["matmul_precision"] = args.matmul_precision + (" (IEEE f32)" if args.matmul_precision == "highest"
                                                                   else " (hipBLASLt f32 emulated with bf16 triples: NOT the line of record)")
    if world > 1:
        import torch.distributed as dist

        assert dist.get_world_size() == args.gpus
    parity_failed = False
    if rank == 0:
        rf = out.get("roofline")
        # what the CPU leg needs goes to the host first, then the GPU is emptied: the --pmc child processes rebuild the
        # graph on the same device (a papers100M-sized share peaks at 93 GB while it is built)
        cpu_args = None
        if want_cpu:
            if kind == "gcn":
                if ctx.get("host_graph") is not None:
                    ei, w = ctx["host_graph"]
                else:
                    pg = ctx["pg"]    # N = 1: rank 0 holds the whole graph (a dry share: its local-source block)
                    ei, w = torch.stack([pg.ei_loc[0], pg.ei_loc[1]]).cpu().contiguous(), pg.w_loc.cpu()
                    del pg
                full = (ei, w, int(out["config"].get("rank0_owned_rows", 0)))
                cpu_args = (args.hidden, sizes_of(args.workload)[3], args.seed, full)
            elif kind == "gat":
                cpu_args = ({"ei": ctx["ei"].cpu(), "n": ctx["n"]}, args.seed)
            else:
                cpu_args = ({"blocks": [type("B", (), {"n_dst_cap": int(b.n_dst_cap)})() for b in ctx["blocks"]],
                             "valid": ctx["valid"]}, args.hidden, args.seed)
        ctx.clear()
        if not emul:
            from gammagl_amd import engine as _eng

            _eng().clear_caches()
            import gc

            gc.collect()
            torch.cuda.empty_cache()
        achievable = None
        if rf and world == 1 and not emul:
            # the yardstick for `frac_of_achievable`: what a 16 B/lane streaming copy (and a 256-byte-row gather) reach
            # on THIS part, timed now with hipEvents (the same launches the --pmc passes calibrate the counters on)
            from gammagl_amd.benchmarks import calibration_launches

            cal_t = calibration_launches(_eng(), dev, time_it=True)
            achievable = cal_t["stream_read"]["GBps"]
            rf["achievable"] = {"stream_read_GBps": cal_t["stream_read"]["GBps"], "stream_copy_GBps": cal_t["stream_copy"]["GBps"],
                                "gather256_GBps": cal_t["gather256"]["GBps"],
                                "note": "bytes moved / hipEvent time (best of 3) of benchmarks.calibration_launches: a 4 GiB "
                                        "16 B/lane streaming read, 2 GiB copied, a 256-byte-row gather through a random "
                                        "permutation; frac_of_achievable uses the streaming READ (the aggregate is ~95 % reads)"}
            if kind not in KERNEL_OF:
                set_traffic(rf, None, "no counter pass for this workload", achievable)
        if rf and world == 1 and not emul and kind in KERNEL_OF:
            t = src = None
            extra = {}
            if args.pmc_traffic != "off":
                t, src, extra = measure_traffic(args, KERNEL_OF[kind], with_l2=args.pmc_traffic == "l2")
            if t is None and kind == "gcn":
                why = src
                t, src = committed_traffic(args, int(rf["launches_per_aggregate"]), int(out["config"].get("rank0_local_edges", -1)))
                if t is not None and why:
                    src += f" [in-run collection unavailable: {why}]"
                elif why:
                    src = f"unavailable: {why}"
            set_traffic(rf, t, src, achievable)
            rf.update(extra)
            if args.pmc_traffic == "l2":   # the other node orders' traffic too (locality workloads)
                for o in out["config"].get("orderings", [])[1:]:
                    t2, src2, extra2 = measure_traffic(args, KERNEL_OF[kind], relabel=o["relabel"], with_l2=True)
                    o["traffic_per_aggregate"] = None if t2 is None else t2 * int(rf["launches_per_aggregate"])
                    o["traffic_over_compulsory"] = None if t2 is None else o["traffic_per_aggregate"] / max(rf["compulsory_bytes"], 1)
                    o["traffic_source"] = src2
                    o.update(extra2)
        if cpu_args is not None:
            out["cpu_baseline"], out["parity"] = \
                {"gcn": cpu_baseline_gcn, "gat": cpu_baseline_gat, "sage": cpu_baseline_sage}[kind](*cpu_args)
        if secondary_wanted(args, world, emul):
            out["secondary"] = run_secondary(args, t_start)
        emit(out)
        if out.get("parity") is not None and not out["parity"]["ok"]:
            # a fast kernel whose results differ from the reference's is not a result: the line above says by how much
            print(f"bench.py: HIP result outside the parity criterion: {out['parity']}", file=sys.stderr, flush=True)
            parity_failed = True
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
