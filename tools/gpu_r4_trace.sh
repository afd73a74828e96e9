#!/bin/bash
# kernel timeline of the products step with the exact-order hub launch (which kernels overlap, how long the hub launch is)
O=gpurun_out/r4c; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b
for side in 1 0; do
GGL_EXACT_SIDE_STREAM=$side timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$side -o t -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_side$side.json 2> $R/$O/bench_side$side.err
cp $(find /tmp/prof_b$side -name '*kernel_stats.csv' | head -1) $R/$O/kernel_stats_side$side.csv
python $R/tools/trace_timeline.py $(find /tmp/prof_b$side -name '*kernel_trace.csv' | head -1) multi_tensor_apply 0.05 > $R/$O/timeline_side$side.txt 2>&1
done
cd $R
python tools/prof_summary.py $O/kernel_stats_side1.csv 12 | cut -c1-170
head -70 $O/timeline_side1.txt | cut -c1-150
