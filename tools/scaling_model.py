#!/usr/bin/env python3
"""Turn gpurun_out/scaling_model.jsonl (tools/scaling_model.sh: one rank's share of the P-way products partition, timed
as a dry partition on one GPU by bench.py --dry-parts P) into the table of profiles/r3_scaling_model.txt: measured
compute per rank + link time MODELLED from the halo volume (floats per halo row per step as the bench line reports them,
over P - 1 xGMI links at `--link-GBps` per direction).

    python tools/scaling_model.py gpurun_out/scaling_model.jsonl [--link-GBps 50] [--one-gpu-ms 81.9]"""
import argparse
import json

p = argparse.ArgumentParser()
p.add_argument("jsonl")
p.add_argument("--link-GBps", type=float, default=50.0)
p.add_argument("--one-gpu-ms", type=float, default=81.9)
p.add_argument("--edges", type=int, default=126167309)
p.add_argument("--aggregations", type=int, default=6)
a = p.parse_args()
rows = [json.loads(line) for line in open(a.jsonl) if line.strip()]
print("products-sized graph, 3-layer GCN h=256 training step, every layer A (X W) (6 aggregations): ONE rank's share of a P-way")
print("partition timed on one MI355X as a dry partition (bench.py --dry-parts P: send lists, send-row gathers, local + halo SpMMs,")
print("reverse scatter, the first layer's halo buffer filled by a GEMM — exactly as in the P-rank run; nothing on the wire).")
print(f"Link time is MODELLED, not measured: halo rows x floats per row per step x 4 B over P - 1 xGMI links at {a.link_GBps:.0f} GB/s per direction.")
print()
print(f"{'P':>2} {'rows/rank':>10} {'edges/rank':>11} {'halo rows':>10} {'floats/row':>10} {'compute ms':>11} {'GB in/step':>11} "
      f"{'GB per link':>12} {'link ms':>8} {'step ms: overlap none..full':>28} {'G edges/s':>16}")
print(f"{1:>2} {2449029:>10} {a.edges:>11} {0:>10} {0:>10} {a.one_gpu_ms:>11.1f} {0:>11} {0:>12} {0:>8} {a.one_gpu_ms:>28.1f} "
      f"{a.aggregations * a.edges / a.one_gpu_ms / 1e6:>16.2f}")
for d in rows:
    P, comp = d["parts"], d["train_step_ms"]
    gb = d["halo_rows"] * d["halo_floats_per_row"] * 4 / 1e9
    per = gb / (P - 1)
    link = per / a.link_GBps * 1e3
    lo, hi = max(comp, link), comp + link
    print(f"{P:>2} {d['owned_rows']:>10} {d['local_edges']:>11} {d['halo_rows']:>10} {d['halo_floats_per_row']:>10} {comp:>11.1f} {gb:>11.2f} {per:>12.2f} "
          f"{link:>8.1f} {f'{hi:.1f} .. {lo:.1f}':>28} {f'{a.aggregations * a.edges / hi / 1e6:.1f} .. {a.aggregations * a.edges / lo / 1e6:.1f}':>16}")
print()
print("(round 2's table modelled the 5-aggregation (A X) W step with 708 floats per halo row: 2.40 GB in per step at P = 8; the input-feature")
print(" halo is now fetched once and layer 1 exchanges nothing: 608 floats per row for the 6-aggregation A (X W) step.)")
