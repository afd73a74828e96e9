#!/usr/bin/env python3
"""Per-op table in the reference's own micro-benchmark protocol (profiler/mpops/complete_test:
ops_cpu/ggl_segment_cpu.py:11-12,24-30 — Ogbn-Arxiv-sized graph, K in {16, 64, 256}, 1 warm-up then timed
repetitions): every op of the path on the MI355X next to the reference's CPU extension (oracle/_ref,
compiled from the reference sources; the oracle C port when it is absent) on the box's host cores.

Lives under tests/ because it executes the checker (oracle/) as the CPU baseline; it is a measurement
script, not a pytest module.      python tests/perf_op_table.py [--cpu-reps 2] > profiles/rN_op_table.txt
"""
import argparse
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def gpu_ms(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def cpu_ms(fn, reps):
    fn()  # the protocol's single warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--widths", default="16,64,256")
    a = ap.parse_args()
    from gammagl_amd import engine
    from gammagl_amd.synth import DATASETS, rmat_graph
    from oracle import oracle as orc

    dev = torch.device("cuda", 0)
    eng = engine()
    ref = None
    try:
        ref = orc.load_ref_ext()
    except Exception:  # noqa: BLE001
        orc.build()
    torch.set_num_threads(1)  # the extension is serial as shipped (setup.py:50 never defines its OpenMP macro)
    n, e, _, _ = DATASETS["arxiv"]
    ei = rmat_graph(n, e, seed=0, device=dev)
    E = ei.shape[1]
    ei_c = ei.cpu()
    g = torch.Generator().manual_seed(0)
    w_c = torch.rand(E, generator=g)
    w = w_c.to(dev)
    dst, dst_c = ei[1].contiguous(), ei_c[1].contiguous()
    print(f"arxiv-sized R-MAT N={n} E={E} (self-loops incl.), f32; GPU = MI355X median of 50 after 5 warm-ups "
          f"(hipEvents); CPU = {'reference _torch_ext (oracle/_ref)' if ref else 'oracle C port'}, 1 thread of "
          f"{os.cpu_count()}, median of {a.cpu_reps} after 1 warm-up")
    print(f"{'op':34s} {'K':>4s} {'GPU ms':>9s} {'Gedges/s':>9s} {'TB/s alg':>9s} {'CPU ms':>10s} {'speed-up':>9s}")
    for K in [int(k) for k in a.widths.split(",")]:
        x_c = torch.randn(n, K, generator=g)
        msg_c = torch.randn(E, K, generator=g)
        x, msg = x_c.to(dev), msg_c.to(dev)
        H = 8 if K >= 64 else 4
        xb_c, wb_c = x_c.view(n, H, K // H), torch.rand(E, H, generator=g)
        xb, wb = xb_c.to(dev), wb_c.to(dev)
        seg_bytes = E * (4 * K + 8) + n * 4 * K
        spmm_bytes = E * (4 * K + 8) + n * (4 * K + 8)
        if ref is not None:
            cpu = {
                "unsorted_segment_sum": lambda: ref.c_segment_sum(msg_c, dst_c, n),
                "unsorted_segment_mean": lambda: ref.c_segment_mean(msg_c, dst_c, n),
                "unsorted_segment_max": lambda: ref.c_segment_max(msg_c, dst_c, n),
                "gspmm sum": lambda: ref.c_spmm_sum(ei_c, w_c, x_c),
                "gspmm mean": lambda: ref.c_spmm_mean(ei_c, w_c, x_c),
                "gspmm max": lambda: ref.c_spmm_max(ei_c, w_c, x_c),
                "bspmm sum": lambda: ref.c_bspmm_sum(ei_c, wb_c, xb_c),
            }
        else:
            ein, wn, xn, mn, dn = ei_c.numpy(), w_c.numpy(), x_c.numpy(), msg_c.numpy(), dst_c.numpy()
            cpu = {
                "unsorted_segment_sum": lambda: orc.segment_sum(mn, dn, n),
                "unsorted_segment_mean": lambda: orc.segment_mean(mn, dn, n),
                "unsorted_segment_max": lambda: orc.segment_max(mn, dn, n),
                "gspmm sum": lambda: orc.spmm_sum_fwd(ein, wn, xn),
                "gspmm mean": lambda: orc.spmm_mean_fwd(ein, wn, xn),
                "gspmm max": lambda: orc.spmm_max_fwd(ein, wn, xn),
                "bspmm sum": lambda: orc.bspmm_sum_fwd(ein, wb_c.numpy(), xb_c.numpy()),
            }
        gpu = {
            "unsorted_segment_sum": (lambda: eng.c_segment_sum(msg, dst, n), seg_bytes),
            "unsorted_segment_mean": (lambda: eng.c_segment_mean(msg, dst, n), seg_bytes),
            "unsorted_segment_max": (lambda: eng.c_segment_max(msg, dst, n), seg_bytes + 8 * n * K),
            "gspmm sum": (lambda: eng.c_spmm_sum(ei, w, x), spmm_bytes),
            "gspmm mean": (lambda: eng.c_spmm_mean(ei, w, x), spmm_bytes),
            "gspmm max": (lambda: eng.c_spmm_max(ei, w, x), spmm_bytes + 8 * n * K),
            "bspmm sum": (lambda: eng.c_bspmm_sum(ei, wb, xb), E * (4 * K + 4 + 4 * H) + n * (4 * K + 8)),
        }
        for name, (fn, nbytes) in gpu.items():
            gms = gpu_ms(fn)
            cms = cpu_ms(cpu[name], a.cpu_reps)
            print(f"{name:34s} {K:4d} {gms:9.3f} {E / gms / 1e6:9.2f} {nbytes / gms / 1e9:9.2f} {cms:10.1f} {cms / gms:8.0f}x",
                  flush=True)
        # the reference's pure-torch formulation (mpops/torch.py:16-18) on every host thread, for the sum
        torch.set_num_threads(os.cpu_count() or 1)
        idx2 = dst_c.view(-1, 1).expand(E, K)
        tms = cpu_ms(lambda: torch.zeros(n, K).scatter_add_(0, idx2, msg_c), a.cpu_reps)
        torch.set_num_threads(1)
        print(f"{'  (torch scatter_add_, all threads)':34s} {K:4d} {'':9s} {'':9s} {'':9s} {tms:10.1f}", flush=True)
        del x, msg, xb, wb


if __name__ == "__main__":
    main()
