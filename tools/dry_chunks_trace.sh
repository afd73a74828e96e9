#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do
rm -rf /tmp/prof_c$c
GGL_HALO_CHUNKS=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -o t -- python $R/bench.py --dry-parts 8 --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_c$c.json 2>/dev/null
python $R/tools/trace_timeline.py $(find /tmp/prof_c$c -name '*kernel_trace.csv' | head -1) multi_tensor_apply 0.04 > $R/$O/timeline_chunks$c.txt 2>&1
done
head -70 $R/$O/timeline_chunks1.txt | cut -c1-120
