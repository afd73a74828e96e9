#!/usr/bin/env python3
"""The WHOLE papers100M-sized graph (N = 111 M, E = 3.23 G > 2^31) aggregated on ONE MI355X: CSR in HBM
(int32 column ids 12.9 GB + f32 weights 12.9 GB + int64 rowptr 0.9 GB), feature panel and result resident
next to it.  The plan comes straight from the CSR (Engine.plan_from_rowptr: no sort, no permutation, so the
2^31 limit of the COO path does not apply).  Graph: in-degrees and source popularity both drawn from the
R-MAT marginals (a Chung-Lu-style stand-in: a joint R-MAT sample would need a 3.2 G-element sort).

    python tools/papers_full_probe.py [--k 64,128,256] [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N_PAPERS, E_PAPERS = 111_059_956, 3_231_371_744


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", default="64,128,256")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from gammagl_amd import engine
    from gammagl_amd.synth import rmat_pairs

    dev = torch.device("cuda", 0)
    eng = engine()
    n, e = int(N_PAPERS * a.scale), int(E_PAPERS * a.scale)
    scale_bits = max(1, (n - 1).bit_length())
    gen = torch.Generator(device=dev).manual_seed(0)
    t0 = time.perf_counter()
    deg = torch.zeros(n, dtype=torch.int64, device=dev)
    col = torch.empty(e, dtype=torch.int32, device=dev)
    filled = 0
    while filled < e:
        m = min(200_000_000, e - filled + (e - filled) // 4 + 1024)
        u, v = rmat_pairs(scale_bits, m, gen, dev)
        ok = (u < n) & (v < n)
        u, v = u[ok][: e - filled], v[ok][: e - filled]
        deg += torch.bincount(v, minlength=n)
        col[filled:filled + u.numel()] = u.to(torch.int32)
        filled += u.numel()
        del u, v, ok
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=rowptr[1:])
    max_deg = int(deg.max())
    del deg
    w = torch.rand(e, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    plan = eng.plan_from_rowptr(rowptr, e, max_len=max_deg)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    res = {"N": n, "E": e, "E_over_2^31": e / 2**31, "max_in_degree": max_deg, "long_rows": plan.n_long,
           "gen_s": round(t_gen, 1), "plan_s": round(t_plan, 3), "runs": []}
    for K in [int(k) for k in a.k.split(",")]:
        free, total = torch.cuda.mem_get_info()
        need = 2 * n * K * 4 + (2 << 30)
        if need > free:
            res["runs"].append({"K": K, "skipped": f"needs {need / 1e9:.0f} GB, {free / 1e9:.0f} GB free"})
            continue
        x = torch.randn(n, K, device=dev)
        out = torch.empty(n, K, device=dev)                  # one result buffer, written in place
        eng.spmm_sum_into(plan, col, w, x, out)              # warm-up + result for the check
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(a.reps):
            eng.spmm_sum_into(plan, col, w, x, out)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.reps
        # spot check: 64 random rows against a direct f64 evaluation
        rows = torch.randint(0, n, (64,), device=dev)
        worst = 0.0
        for r in rows.tolist():
            b, t = int(rowptr[r]), int(rowptr[r + 1])
            ref = (w[b:t].double().unsqueeze(1) * x[col[b:t].long()].double()).sum(0)
            bound = (w[b:t].double().unsqueeze(1) * x[col[b:t].long()].double().abs()).sum(0)
            worst = max(worst, float(((out[r].double() - ref).abs() / bound.clamp_min(1e-30)).max()))
        assert worst < 1e-5, worst
        alg = e * (4 * K + 8) + n * (4 * K + 8)
        res["runs"].append({"K": K, "ms": round(ms, 2), "Gedges_per_s": round(e / ms / 1e6, 2),
                            "TBps_alg": round(alg / ms / 1e9, 2), "rel_err_64_rows": worst,
                            "hbm_in_use_GB": round(torch.cuda.memory_allocated() / 1e9, 1)})
        print(json.dumps(res["runs"][-1]), flush=True)
        del x, out
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
