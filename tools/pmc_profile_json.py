#!/usr/bin/env python3
"""Assemble profiles/r2_pmc_products_k<K>.json from the per-kernel counter summary tools/pmc_kernel.sh wrote
(gpurun_out/pmc_<wl>_k<K>.json): the SpMM-sum kernel's HBM bytes per LAUNCH beside its algorithmic bytes per launch
(a K-wide aggregate is `launches` launches over K / launches columns each — reduce.hip launch_f32_cols).

    python tools/pmc_profile_json.py gpurun_out/pmc_products_k256.json 256 126167309 2449029 profiles/r2_pmc_products_k256.json"""
import json
import sys

src, K, E, N, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
rec = json.load(open(src))
name = [k for k in rec if "row_reduce_kernel<float, 4, 0, 1, 1," in k and "hbm_bytes_per_launch" in rec[k]]
assert len(name) == 1, name
k = rec[name[0]]
calls = int(k["FETCH_SIZE"]["dispatches"])
aggregates = 4                                     # tools/pmc_probe.py launches the aggregate 4 times
launches = calls // aggregates
Kl = K // launches
out = {
    "graph": "products-sized R-MAT of bench.py (synth.rmat_partitioned, seed 0, random relabel, order src)",
    "graph_edges": E, "graph_nodes": N, "K": K, "launches_per_aggregate": launches, "K_per_launch": Kl,
    "kernel": name[0],
    "collected": "rocprofv3 --pmc, one pass per counter set (tools/pmc_kernel.sh): FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum "
                 "TCC_MISS_sum; KiB units; gfx950 read-side x2 correction (MI355X_MICROARCH.md, HBM section); averages per launch",
    "counters": {c: k[c] for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum") if c in k},
    "l2_hit_rate": k.get("l2_hit_rate"),
    f"spmm_sum_k{K}": {
        "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
        "alg_bytes_per_launch": E * (4 * Kl + 8) + N * (4 * Kl + 8),
        "l2_miss_lines_x_128B": k["TCC_MISS_sum"]["avg"] * 128.0 if "TCC_MISS_sum" in k else None,
        "hbm_bytes_per_aggregate": k["hbm_bytes_per_launch"] * launches,
    },
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out[f"spmm_sum_k{K}"]), "launches", launches, "l2 hit", out["l2_hit_rate"])
