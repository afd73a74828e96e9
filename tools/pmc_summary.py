#!/usr/bin/env python3
"""Average PMC counter value per dispatch, per (short) kernel name, from rocprofv3 counter CSVs.

usage: pmc_summary.py out.json fetch_counter_collection.csv write_counter_collection.csv [l2 csv]
HBM bytes per launch follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced read stream, which is what
these kernels issue, so the read side is doubled: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    merged = defaultdict(dict)
    for p in paths:
        for k, counters in load(p).items():
            if not k.startswith("ggl::"):
                continue
            for c, vals in counters.items():
                merged[k][c] = {"avg": sum(vals) / len(vals), "dispatches": len(vals)}
    res = {}
    for k, c in merged.items():
        e = dict(c)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"]["avg"] + c["WRITE_SIZE"]["avg"]) * 1024.0
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            h, m = c["TCC_HIT_sum"]["avg"], c["TCC_MISS_sum"]["avg"]
            e["l2_hit_rate"] = h / (h + m) if h + m > 0 else None
        res[k] = e
    json.dump(res, open(out_path, "w"), indent=1)
    for k, e in res.items():
        print(k[:90], {a: (round(b, 4) if isinstance(b, float) else b["avg"]) for a, b in e.items()})


if __name__ == "__main__":
    main()
