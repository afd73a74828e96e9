#!/bin/bash
# step-level timeline of the default bench step (what is NOT aggregation)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o r2 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison > $R/$O/h_bench.json 2> $R/$O/h_bench.err
cd $R
python tools/trace_timeline.py $(find /tmp/prof_h -name '*kernel_trace.csv' | head -1) > $O/h_timeline.txt 2>&1
cp $(find /tmp/prof_h -name '*kernel_stats.csv' | head -1) $O/h_kernel_stats.csv
cat $O/h_timeline.txt | cut -c1-150
