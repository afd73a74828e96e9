#!/usr/bin/env python3
"""Compact view of a rocprofv3 *_kernel_stats.csv (short kernel names, ms, calls, share)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("at::native::", "")
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT\d+x\d+x\d+", name)
        return "hipBLASLt GEMM " + (m.group(0) if m else "")
    return name[:100]


def main(path, top=30):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
    for r in rows[:top]:
        print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {int(r['Calls']):5d}x avg {float(r['AverageNs']) / 1e6:8.3f} ms "
              f"{float(r['Percentage']):5.1f}%  {short(r['Name'])}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
