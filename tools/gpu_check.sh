#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; rm -rf $O/prof_* ; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1200 bash -c "python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1"
run smoke   timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run products timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run rocprof timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_bench.json 2> $R/$O/rocprof.err"
find $O -name "*.db" -delete; find $O -type f -size +20M -delete
cat $O/summary.txt; tail -4 $O/pytest_gpu.log | cut -c1-300; tail -2 $O/smoke.log; cut -c1-900 $O/bench_products.json; tail -3 $O/bench_products.err
python tools/prof_summary.py $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) 22
