#!/bin/bash
# the two commands the driver runs on the GPU box at round end, plus the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/check; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu.log | cut -c1-200
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
