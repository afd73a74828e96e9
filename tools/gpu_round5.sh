#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; rm -rf $O/prof_* ; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1200 bash -c "python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1"
run configs timeout 1500 bash -c "python tools/bench_configs.py 2 > $O/bench_configs.log 2>&1"
cat $O/summary.txt; tail -15 $O/pytest_gpu.log | cut -c1-300; cat $O/bench_configs.log | cut -c1-400
