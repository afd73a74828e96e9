#!/bin/bash
# Round 3, second GPU pass: suites at HEAD, the sorted-plan bspmm weight gradient (A/B), the row hand-out order (A/B),
# the dense-block MFMA prototype, the papers100M-sized share as a bench workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3b; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 900 bash -c "python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1"
run bspmm    timeout 600 bash -c "python tools/bspmm_bwd_probe.py $O/bspmm_bwd.txt > $O/bspmm.log 2>&1"
run roword   timeout 900 bash -c "python tools/roworder_probe.py $O/roworder.txt > $O/roworder.log 2>&1"
run mfma     timeout 600 bash -c "python tools/dense_block_mfma_probe.py $O/dense_block_mfma.txt > $O/mfma.log 2>&1"
run share    timeout 900 bash -c "python bench.py --workload papers-share --steps 3 --warmup 1 > $O/bench_papers_share.json 2> $O/bench_papers_share.err"
cat $O/summary.txt; tail -12 $O/pytest_gpu.log | cut -c1-220
cat $O/bspmm_bwd.txt $O/roworder.txt $O/dense_block_mfma.txt 2>/dev/null; tail -3 $O/bspmm.log $O/roworder.log $O/mfma.log | cut -c1-300
tail -c 1200 $O/bench_papers_share.err | tail -4; head -c 3000 $O/bench_papers_share.json
