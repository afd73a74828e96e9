#!/usr/bin/env python3
"""Timeline view of a rocprofv3 kernel trace: per queue, the launches of the LAST training step (between
the last two Adam kernels' neighbourhoods) with start offset, duration and idle gap — shows what sits
on the critical path and what overlaps.   python tools/trace_timeline.py <kernel_trace.csv> [marker] [min_ms]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("at::native::", "")
    if n.startswith("Cijk_"):
        m = re.search(r"MT\d+x\d+x\d+", n)
        return "hipBLASLt " + (m.group(0) if m else "")
    return n[:70]


rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marker = sys.argv[2] if len(sys.argv) > 2 else "multi_tensor_apply"
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15   # launches shorter than this (and not after a gap) are not listed
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# a step = from the first kernel after the previous step's last optimizer kernel to this step's last one
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] - marks[i] > 20]
if len(ends) < 2:
    sys.exit("need two steps in the trace")
lo, hi = ends[-2] + 1, ends[-1]
t0 = rows[lo]["s"]
print(f"last step: {hi - lo + 1} launches, {(rows[hi]['e'] - t0) / 1e6:.2f} ms wall")
busy_until = {}
for r in rows[lo:hi + 1]:
    q = r["Queue_Id"]
    gap = (r["s"] - busy_until.get(q, r["s"])) / 1e6
    busy_until[q] = r["e"]
    dur = (r["e"] - r["s"]) / 1e6
    if dur >= min_ms or gap >= 0.15:
        print(f"q{q} +{(r['s'] - t0) / 1e6:8.2f} ms  dur {dur:7.3f}  gap {gap:6.3f}  {short(r['Kernel_Name'])}")
