#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (the A (X W) step only: --no-comparison) + the step's timeline;
# the op sweep at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final3; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison > $R/$O/rocprof_bench.json 2> $R/$O/rocprof_bench.err
cp $(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $R/$O/bench_kernel_stats.csv
python $R/tools/trace_timeline.py $(find /tmp/prof_b -name '*kernel_trace.csv' | head -1) > $R/$O/bench_timeline.txt 2>&1
cd $R
python tools/prof_summary.py $O/bench_kernel_stats.csv 18 > $O/bench_summary.txt; head -12 $O/bench_summary.txt | cut -c1-160
head -60 $O/bench_timeline.txt | cut -c1-150
timeout 600 python tools/ops_shape_sweep.py products > $O/ops_shape_sweep_products.txt 2>&1; tail -42 $O/ops_shape_sweep_products.txt | head -20
