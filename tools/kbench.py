#!/usr/bin/env python3
"""tools/kbench.py — within-process A/B micro-benchmark of the hot kernels on one MI355X.

Interleaved rounds (guide §5.4 rule 24), hipEvent timing on the launch stream via
ggl_time_spmm_sum for the dominant kernel and torch.cuda events for the others (all launches go to
torch's current stream).  Prints one line per variant: median ms, edges/s, algorithmic GB/s
(E*(4K+8) + N*(4K+8) for the fused SpMM; E*(4K+8) + N*4K for a segment op).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev_time(fn, reps):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-layout", action="store_true")
    args = ap.parse_args()
    from gammagl_amd import engine
    from gammagl_amd.layers import calc_gcn_norm
    from gammagl_amd.synth import DATASETS, rmat_graph

    dev = torch.device("cuda", 0)
    eng = engine()
    n, e, _, _ = DATASETS[args.workload]
    results = []

    def report(name, ms_list, E, N, K, kind="spmm"):
        ms = statistics.median(ms_list)
        b = E * (4 * K + 8) + N * (4 * K + 8) if kind == "spmm" else E * (4 * K + 8) + N * 4 * K
        r = {"name": name, "ms_median": ms, "ms_min": min(ms_list), "gedges_s": E / ms / 1e6,
             "alg_GBs": b / ms / 1e6, "K": K, "E": E, "N": N}
        results.append(r)
        print(f"{name:58s} K={K:4d} {ms:9.3f} ms (min {min(ms_list):8.3f})  {E / ms / 1e6:7.2f} Gedges/s  "
              f"{b / ms / 1e6:8.1f} GB/s alg  ({100 * b / ms / 1e6 / 8000:5.1f}% of 8 TB/s)", flush=True)

    layouts = [("src", "random")] if args.skip_layout else [("src", "random"), ("dst", "random"), ("src", "degree")]
    g = torch.Generator(device=dev).manual_seed(0)
    for order, relabel in layouts:
        t0 = time.perf_counter()
        ei = rmat_graph(n, e, seed=0, device=dev, order=order, relabel=relabel)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        E = ei.shape[1]
        t0 = time.perf_counter()
        gp = eng.graph_plan(ei, n)
        gp.bwd  # noqa: B018
        torch.cuda.synchronize()
        t_plan = time.perf_counter() - t0
        deg = gp.fwd.counts()
        print(f"== {args.workload} order={order} relabel={relabel}: N={n} E={E} gen {t_gen:.1f}s plan(fwd+bwd) {t_plan:.2f}s "
              f"max_deg={int(deg.max())} long_rows={gp.fwd.n_long} chunks={gp.fwd.n_chunks} sorted={gp.fwd.is_sorted}", flush=True)
        w = calc_gcn_norm(ei, n).contiguous()
        for K in (256, 64, 47, 16):
            x = torch.randn(n, K, generator=g, device=dev)
            # (u4 swz0) is the shipped default; the XCD remap (swz1) and deeper unroll are A/B variants
            variants = [("u4 swz0", 4, 0), ("u8 swz0", 8, 0), ("u4 swz1", 4, 1)] if K == 256 else [("u4 swz0", 4, 0)]
            acc = {v[0]: [] for v in variants}
            accT = []
            for _ in range(args.rounds):
                for name, u, sw in variants:
                    eng.set_option("unroll", u)
                    eng.set_option("xcd_swizzle", sw)
                    acc[name].append(eng.time_spmm_sum(gp, w, x, reps=args.reps))
                eng.set_option("unroll", 4)
                eng.set_option("xcd_swizzle", 0)
                accT.append(ev_time(lambda: eng._spmm_fwd("sum", gp.bwd, gp.colT, w, x, n), args.reps))
            for name, _, _ in variants:
                report(f"spmm_sum fwd [{order}/{relabel}] {name}", acc[name], E, n, K)
            report(f"spmm_sum transposed (bwd) [{order}/{relabel}]", accT, E, n, K)
            if K == 256:
                ms_now = [ev_time(lambda: eng._spmm_fwd("sum", gp.fwd, gp.col, None, x, n), args.reps) for _ in range(args.rounds)]
                report(f"spmm_sum fwd, no weights [{order}/{relabel}]", ms_now, E, n, K)
                ms_max = [ev_time(lambda: eng._spmm_fwd("max", gp.fwd, gp.col, w, x, n), args.reps) for _ in range(args.rounds)]
                report(f"spmm_max fwd [{order}/{relabel}]", ms_max, E, n, K)
        # K = 1 degree-style segment_sum on both id vectors
        ones = torch.ones(E, device=dev)
        for nm, ids in (("dst", ei[1]), ("src", ei[0])):
            ms1 = [ev_time(lambda: eng.c_segment_sum(ones, ids, n), args.reps) for _ in range(args.rounds)]
            report(f"segment_sum K=1 (degree) over {nm} [{order}/{relabel}]", ms1, E, n, 1, kind="seg")
        del ei, gp, w
        eng.seg_cache.clear()
        eng.graph_cache.clear()
        torch.cuda.empty_cache()

    # unfused segment ops on a pre-gathered message tensor (arxiv size, profiler protocol K = 16/64/256)
    na, ea, _, _ = DATASETS["arxiv"]
    ei = rmat_graph(na, ea, seed=0, device=dev)
    E = ei.shape[1]
    dst = ei[1].contiguous()
    for K in (16, 64, 256):
        x = torch.randn(na, K, generator=g, device=dev)
        msg = x[ei[0]]
        for nm, fn in (("sum", eng.c_segment_sum), ("mean", eng.c_segment_mean), ("max", eng.c_segment_max)):
            ms = [ev_time(lambda: fn(msg, dst, na), args.reps) for _ in range(args.rounds)]
            report(f"arxiv unsorted_segment_{nm} on [E,K] messages", ms, E, na, K, kind="seg")
        ms = [ev_time(lambda: eng.c_spmm_sum(ei, None, x), args.reps) for _ in range(args.rounds)]
        report("arxiv gspmm sum (fused)", ms, E, na, K)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
