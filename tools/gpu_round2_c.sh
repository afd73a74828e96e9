#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1"
run sage    timeout 600 bash -c "python tools/sage_minibatch_probe.py > $O/sage_minibatch.txt 2>&1"
run narrowA timeout 600 bash -c "python tools/narrow_probe.py arxiv > $O/narrow_arxiv.txt 2>&1"
run narrowP timeout 600 bash -c "python tools/narrow_probe.py products > $O/narrow_products.txt 2>&1"
run bench   timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run rocprofB timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/rocprof_bench.json 2> $R/$O/rocprof_bench.err; cp \$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $R/$O/r2_bench_kernel_stats.csv"
run rocprofG timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_g && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o r2 -- python $R/tools/gat_profile.py > $R/$O/rocprof_gat.log 2>&1; cp \$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1) $R/$O/r2_gat_kernel_stats.csv"
cat $O/summary.txt; tail -8 $O/pytest_gpu.log | cut -c1-300
grep -v "amdgpu.ids" $O/sage_minibatch.txt; grep -v "amdgpu.ids" $O/narrow_arxiv.txt | cut -c1-250; grep -v "amdgpu.ids" $O/narrow_products.txt | cut -c1-250
cut -c1-400 $O/bench_products.json; echo; cut -c1-300 $O/rocprof_bench.json; echo
python tools/prof_summary.py $O/r2_bench_kernel_stats.csv 14; python tools/prof_summary.py $O/r2_gat_kernel_stats.csv 12
