#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_p
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench.json 2>/dev/null
python $R/tools/trace_timeline.py $(find /tmp/prof_p -name '*kernel_trace.csv' | head -1) multi_tensor_apply 0.05 > $R/$O/timeline.txt 2>&1
cp $(find /tmp/prof_p -name '*kernel_stats.csv' | head -1) $R/$O/kernel_stats.csv
head -30 $R/$O/timeline.txt | cut -c1-125
