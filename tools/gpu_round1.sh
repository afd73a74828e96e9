#!/bin/bash
# tools/gpu_round1.sh — one gpurun call: smoke, GPU parity suite, bench (arxiv + products), kernel A/B,
# rocprofv3 kernel trace.  Everything lands under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run smoke   timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run pytest  timeout 1200 bash -c "python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1"
run arxiv   timeout 300 bash -c "python bench.py --workload arxiv --steps 10 --warmup 3 > $O/bench_arxiv.json 2> $O/bench_arxiv.err"
run products timeout 900 bash -c "python bench.py --steps 5 --warmup 2 > $O/bench_products.json 2> $O/bench_products.err"
run kbench  timeout 1200 bash -c "python tools/kbench.py --rounds 3 --reps 2 --out $O/kbench_r1.json > $O/kbench.log 2>&1"
R=$PWD
run rocprof timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_products -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_bench.json 2> $R/$O/rocprof.err"
cat $O/summary.txt; tail -5 $O/smoke.log; tail -15 $O/pytest_gpu.log; cat $O/bench_arxiv.json; tail -3 $O/bench_arxiv.err; cat $O/bench_products.json; tail -3 $O/bench_products.err; tail -60 $O/kbench.log
find $O/prof_products -name "*stats*" | head
