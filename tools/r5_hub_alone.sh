#!/bin/bash
# the hub walk alone vs beside the row walks: per-kernel durations of tools/r5_hub_alone_probe.py (rocprofv3 --kernel-trace)
O=gpurun_out/${1:-r5f}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/ph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -o t -- python tools/r5_hub_alone_probe.py 2>&1 | grep -v "rocprofv3\|amdgpu\|HSA version\|output_stream" > $O/hub_alone.txt
t=$(find /tmp/ph -name "*kernel_trace.csv" | head -1)
python - "$t" >> $O/hub_alone.txt <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hub_rows" in r["Kernel_Name"] or "row_reduce" in r["Kernel_Name"] or "long_final" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    kind = "hub  " if "hub_rows" in r["Kernel_Name"] else "final" if "long_final" in r["Kernel_Name"] else "row  "
    print(f"{kind} start {(s - t0) / 1e6:9.3f}  dur {(e - s) / 1e6:7.3f}  end {(e - t0) / 1e6:9.3f}")
PY
head -12 $O/hub_alone.txt
