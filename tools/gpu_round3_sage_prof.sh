#!/bin/bash
# kernel trace of the replayed GraphSAGE mini-batch step (bench.py --workload sage-minibatch): which of its ~110 small
# kernels the 0.7 ms go to
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/sage; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $R/bench.py --workload sage-minibatch --steps 50 --warmup 5 --no-cpu-baseline --pmc-traffic off > $R/$O/bench.json 2> $R/$O/bench.err
cp $(find /tmp/prof_s -name '*kernel_stats.csv' | head -1) $R/$O/kernel_stats.csv
python $R/tools/trace_timeline.py $(find /tmp/prof_s -name '*kernel_trace.csv' | head -1) multi_tensor_apply 0 > $R/$O/timeline.txt 2>&1
cd $R
python tools/prof_summary.py $O/kernel_stats.csv 30 > $O/summary.txt; head -34 $O/summary.txt | cut -c1-170
head -130 $O/timeline.txt | cut -c1-140
