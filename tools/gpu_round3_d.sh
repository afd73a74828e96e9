#!/bin/bash
# Round 3, fourth GPU pass: suites at HEAD (CPU key, capture with collectives, new dropout words), pipelined bspmm
# weight gradient, XCD run-length swizzle A/B, where the time of a 16-bit K=47 sum goes, halo statistics of the new
# planted graph, GAT bench (new dropout generator).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3d; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 900 bash -c "python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1"
run bspmm    timeout 600 bash -c "python tools/bspmm_bwd_probe.py $O/bspmm_bwd.txt > $O/bspmm.log 2>&1"
run roword   timeout 900 bash -c "python tools/roworder_probe.py $O/roworder.txt > $O/roworder.log 2>&1"
for K in 47 16; do
run half$K   timeout 600 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h$K && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h$K -o h -- python $R/tools/half_one.py $K > $R/$O/half_one_$K.log 2>&1; python $R/tools/prof_summary.py \$(find /tmp/prof_h$K -name '*kernel_stats.csv' | head -1) 12 > $R/$O/half_profile_K$K.txt"
done
run halo     timeout 900 bash -c "python tools/halo_stats.py products > $O/halo_stats.txt 2>&1"
run gat      timeout 600 bash -c "python bench.py --workload reddit-gat --no-cpu-baseline > $O/bench_reddit_gat.json 2> $O/bench_reddit_gat.err"
cat $O/summary.txt; tail -8 $O/pytest_gpu.log | cut -c1-220
cat $O/bspmm_bwd.txt $O/roworder.txt $O/half_profile_K47.txt $O/half_profile_K16.txt $O/halo_stats.txt 2>/dev/null | cut -c1-260
tail -n 2 $O/bspmm.log $O/roworder.log | cut -c1-300
python -c "
import json; d=json.load(open('$O/bench_reddit_gat.json')); print('gat', round(d['ms_per_step'],2), d['config']['fused_gat_forward_ms'], d['roofline']['frac'])"
