#!/usr/bin/env python3
"""Turn gpurun_out/scaling_model.jsonl (tools/scaling_model.sh: one rank's share of the P-way products partition, timed
as a dry partition on one GPU) into the table of profiles/r2_scaling_model.txt: measured compute per rank + link time
MODELLED from the halo volume (708 floats per halo row per step over P - 1 xGMI links at `--link-GBps` per direction).

    python tools/scaling_model.py gpurun_out/scaling_model.jsonl [--link-GBps 50] [--one-gpu-ms 62.6]"""
import argparse
import json

p = argparse.ArgumentParser()
p.add_argument("jsonl")
p.add_argument("--link-GBps", type=float, default=50.0)
p.add_argument("--one-gpu-ms", type=float, default=62.6)
p.add_argument("--edges", type=int, default=126167309)
a = p.parse_args()
rows = [json.loads(line) for line in open(a.jsonl) if line.strip()]
floats_per_halo_row = 100 + 256 + 256 + 48 + 48      # layer 1 forward only; layer 2 and the class layer both ways
print(f"{'P':>2} {'rows/rank':>10} {'edges/rank':>11} {'halo rows':>10} {'compute ms':>11} {'GB in/step':>11} "
      f"{'GB per link':>12} {'link ms':>8} {'step ms: overlap none..full':>28} {'G edges/s':>16}")
print(f"{1:>2} {2449029:>10} {a.edges:>11} {0:>10} {a.one_gpu_ms:>11.1f} {0:>11} {0:>12} {0:>8} {a.one_gpu_ms:>28.1f} "
      f"{5 * a.edges / a.one_gpu_ms / 1e6:>16.2f}")
for d in rows:
    P, comp = d["parts"], d["train_step_ms"]
    gb = d["halo_rows"] * floats_per_halo_row * 4 / 1e9
    per = gb / (P - 1)
    link = per / a.link_GBps * 1e3
    lo, hi = max(comp, link), comp + link
    print(f"{P:>2} {d['owned_rows']:>10} {d['local_edges']:>11} {d['halo_rows']:>10} {comp:>11.1f} {gb:>11.2f} {per:>12.2f} "
          f"{link:>8.1f} {f'{hi:.1f} .. {lo:.1f}':>28} {f'{5 * a.edges / hi / 1e6:.1f} .. {5 * a.edges / lo / 1e6:.1f}':>16}")
