#!/bin/bash
# Round 3, first GPU pass: suites (incl. the papers100M-sized share test), the reworked bench line, the new bench
# workloads, the dense-block (MFMA) probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3a; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 900 bash -c "python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1"
run smoke    timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run bench    timeout 600 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run dense    timeout 300 bash -c "python tools/dense_block_probe.py products $O/dense_block_probe.txt > $O/dense.log 2>&1"
run share    timeout 600 bash -c "python bench.py --workload papers-share --steps 3 --warmup 1 > $O/bench_papers_share.json 2> $O/bench_papers_share.err"
run planted  timeout 600 bash -c "python bench.py --workload products-planted --pmc-traffic l2 --no-cpu-baseline > $O/bench_planted.json 2> $O/bench_planted.err"
run gat      timeout 600 bash -c "python bench.py --workload reddit-gat > $O/bench_reddit_gat.json 2> $O/bench_reddit_gat.err"
run sage     timeout 600 bash -c "python bench.py --workload sage-minibatch --steps 200 --warmup 20 > $O/bench_sage.json 2> $O/bench_sage.err"
cat $O/summary.txt; tail -15 $O/pytest_gpu.log | cut -c1-220; tail -1 $O/smoke.log
for f in products papers_share planted reddit_gat sage; do echo "== $f"; tail -c 1500 $O/bench_$f.err | tail -5; head -c 600 $O/bench_$f.json; echo; done
cat $O/dense_block_probe.txt
