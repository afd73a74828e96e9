#!/bin/bash
# first GPU pass of round 2: tests, smoke, bench, GAT fast-vs-generic, narrow widths, halo overhead
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1"
run smoke   timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run bench   timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run gat88   timeout 600 bash -c "python tools/gat_probe2.py 8 8 > $O/gat_probe_8x8.txt 2>&1"
run narrowA timeout 600 bash -c "python tools/narrow_probe.py arxiv > $O/narrow_arxiv.txt 2>&1"
run narrowP timeout 600 bash -c "python tools/narrow_probe.py products > $O/narrow_products.txt 2>&1"
run halo    timeout 900 bash -c "python tools/self_halo_probe.py products > $O/self_halo.txt 2>&1"
cat $O/summary.txt; tail -25 $O/pytest_gpu.log | cut -c1-300; tail -2 $O/smoke.log; cut -c1-1500 $O/bench_products.json; tail -3 $O/bench_products.err
cat $O/gat_probe_8x8.txt; cat $O/narrow_arxiv.txt; cat $O/narrow_products.txt; tail -5 $O/self_halo.txt
