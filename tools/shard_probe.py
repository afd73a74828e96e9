#!/usr/bin/env python3
"""One GPU's share of BASELINE config 5 (3-layer GCN, ogbn-papers100M-sized graph, 8 GPUs) on ONE MI355X.

The 8-GPU job itself cannot be launched from this session; what one rank executes per aggregate can:
rows = N/8 destination nodes, E/8 in-edges (+ loops), sources spread over local + halo slots
(R-MAT popularity, random node partition = the worst case for the halo: most touched sources are
remote).  The probe builds that rectangular block, times the plan build, the forward SpMM-sum
(K = 256) and the transposed one the backward pass runs, and checks the result with a
size-independent property: column sums of the output == (per-source weight sums) @ x.

    python tools/shard_probe.py [--parts 8] [--k 256] [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

N_PAPERS, E_PAPERS = 111_059_956, 3_231_371_744   # SURVEY.md §8 sizes (symmetrised directed edges)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink factor for a quick run")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from gammagl_amd import engine
    from gammagl_amd.synth import rmat_pairs

    dev = torch.device("cuda", 0)
    eng = engine()
    n_all = int(N_PAPERS * a.scale)
    n_dst = n_all // a.parts
    e_loc = int(E_PAPERS * a.scale) // a.parts
    gen = torch.Generator(device=dev).manual_seed(0)
    sc_all = max(1, (n_all - 1).bit_length())
    t0 = time.perf_counter()
    src_l, dst_l = [], []
    have = 0
    while have < e_loc:
        m = min(e_loc - have + (e_loc - have) // 2 + 1024, 200_000_000)
        u, v = rmat_pairs(sc_all, m, gen, dev)
        ok = (u < n_all) & (v < n_all)
        u, v = u[ok], v[ok]
        src_l.append(u)
        dst_l.append(v)
        have += u.numel()
        del ok
    gsrc = torch.cat(src_l)[:e_loc]
    gdst = torch.cat(dst_l)[:e_loc]
    del src_l, dst_l, u, v
    # random node partition: a node's owner/local id come from a random relabelling; this rank owns
    # relabelled ids [0, n_dst).  Destinations are folded onto the owned range (every edge of the block
    # ends in an owned row by construction of a 1-D row partition); sources keep their global identity.
    pi = torch.randperm(n_all, generator=gen, device=dev)
    gdst = pi[gdst] % n_dst
    gsrc = pi[gsrc]
    del pi
    # slots: owned sources keep their local id, every distinct remote source gets one halo slot
    # (owned ids 0..n_dst-1 sort first, so slot == local id for them; the appended arange is also the
    # source side of the self-loops add_self_loops puts on the owned rows)
    loops = torch.arange(n_dst, dtype=torch.int64, device=dev)
    uniq, src = torch.unique(torch.cat([gsrc, loops]), return_inverse=True)
    n_src = int(uniq.numel())
    n_halo = n_src - n_dst
    del gsrc, uniq
    dst = torch.cat([gdst, loops])
    del gdst, loops
    index = torch.stack([src, dst]).contiguous()
    del src, dst
    E = index.shape[1]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    w = torch.rand(E, device=dev)
    x = torch.randn(n_src, a.k, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp = eng.graph_plan(index, n_dst, n_src)
    gp.bwd, gp.colT  # noqa: B018  (build the transposed plan too: the backward pass needs it)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0

    ms_f = eng.time_spmm_sum(gp, w, x, reps=a.reps)
    out = eng.spmm(gp, w, x)
    # property: sum_i out[i,:] == sum_j (sum_{e: src=j} w_e) x[j,:]
    cw = torch.zeros(n_src, device=dev, dtype=torch.float64).index_add_(0, index[0], w.double())
    got = out.sum(0, dtype=torch.float64)
    want = torch.zeros(a.k, dtype=torch.float64, device=dev)
    bound = torch.zeros(a.k, dtype=torch.float64, device=dev)
    step = max(1, n_src // 16)
    for r0 in range(0, n_src, step):   # chunked so no second [n_src, K] temporary is needed
        xc, cc = x[r0:r0 + step], cw[r0:r0 + step].float()
        want += (cc @ xc).double()
        bound += (cc.abs() @ xc.abs()).double()
    rel = float(((got - want).abs() / bound.clamp_min(1e-30)).max().item())
    del cw, xc, cc
    # transposed aggregate (what backward runs): out_T [n_src, K] from g [n_dst, K]
    g = out
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):  # warm-up: sorted weights cached, the allocator holds both output buffers
        gx, _ = eng._spmm_fwd("sum", gp.bwd, gp.colT, w, g, gp.N_src)
    ev0.record()
    for _ in range(a.reps):
        gx, _ = eng._spmm_fwd("sum", gp.bwd, gp.colT, w, g, gp.N_src)
    ev1.record()
    torch.cuda.synchronize()
    ms_b = ev0.elapsed_time(ev1) / a.reps
    bytes_f = E * (4 * a.k + 8) + n_dst * (4 * a.k + 8)
    bytes_b = E * (4 * a.k + 8) + n_src * (4 * a.k + 8)
    res = {
        "workload": f"papers100M/{a.parts} shard (scale {a.scale})", "rows": n_dst, "edges": E,
        "src_slots": n_src, "halo_slots": n_halo, "halo_GB": n_halo * a.k * 4 / 1e9, "K": a.k,
        "gen_s": t_gen, "plan_build_s": t_plan,
        "spmm_fwd_ms": ms_f, "spmm_fwd_TBps_alg": bytes_f / ms_f / 1e9,
        "spmm_bwd_ms": ms_b, "spmm_bwd_TBps_alg": bytes_b / ms_b / 1e9,
        "edges_per_s_fwd": E / ms_f * 1e3, "colsum_rel_err": rel,
        "hbm_peak_GB": torch.cuda.max_memory_allocated() / 1e9,
    }
    assert rel < 1e-5, rel
    print(json.dumps(res))


if __name__ == "__main__":
    main()
