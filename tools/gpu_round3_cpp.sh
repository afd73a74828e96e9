#!/bin/bash
# GPU pass for the C++-registered ops: the driver's two commands, then the per-call host cost of the three routes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/cpp; mkdir -p $O
: # (tools/gpu_check.sh: removed in round 5)
timeout 600 python tools/op_overhead_probe.py > $O/op_overhead.txt 2>&1; tail -5 $O/op_overhead.txt
timeout 600 python tools/op_overhead_probe.py --nodes 169343 --edges 2315598 --width 256 --reps 300 > $O/op_overhead_arxiv.txt 2>&1; tail -4 $O/op_overhead_arxiv.txt
