#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1"
run lastlayer timeout 900 bash -c "python tools/gat_lastlayer_probe.py > $O/gat_lastlayer.txt 2>&1"
run sage    timeout 600 bash -c "python tools/sage_minibatch_probe.py > $O/sage_minibatch.txt 2>&1"
cat $O/summary.txt; tail -30 $O/pytest_gpu.log | cut -c1-400
grep -v "amdgpu.ids" $O/gat_lastlayer.txt | tail -12; grep -v "amdgpu.ids" $O/sage_minibatch.txt | tail -9
