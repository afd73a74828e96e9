#!/usr/bin/env python3
"""SpMM-sum (K = 256) throughput as a function of row length: uniform rows of 1..64 elements with
random sources, the regime the transposed halo block of a partitioned graph lives in
(tools/shard_probe.py).  Prints one line per configuration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from gammagl_amd import engine

    eng = engine()
    dev = torch.device("cuda", 0)
    K = int(os.environ.get("K", "256"))
    e_total = int(os.environ.get("E", str(64_000_000)))
    n_src = int(os.environ.get("NSRC", str(8_000_000)))
    x = torch.randn(n_src, K, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    for length in (1, 2, 4, 8, 16, 32, 64):
        n_rows = e_total // length
        E = n_rows * length
        dst = torch.arange(n_rows, device=dev).repeat_interleave(length)
        src = torch.randint(0, n_src, (E,), generator=g, device=dev)
        index = torch.stack([src, dst])
        w = torch.rand(E, device=dev)
        for ro in (1, 2):   # 1 = natural row order for wave-per-row kernels, 2 = longest-first order
            eng.set_option("row_order", ro)
            gp = eng.graph_plan(index.clone(), n_rows, n_src)
            ms = eng.time_spmm_sum(gp, w, x, reps=5)
            alg = E * (4 * K + 8) + n_rows * (4 * K + 8)
            print(f"len={length:3d} rows={n_rows:9d} row_order={ro} ms={ms:8.3f} alg_TBps={alg / ms / 1e9:6.2f} "
                  f"rows_per_us={n_rows / ms / 1e3:8.1f}", flush=True)
        del index, dst, src, w, gp
    eng.set_option("row_order", 1)


if __name__ == "__main__":
    main()
