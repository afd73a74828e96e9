#!/bin/bash
O=gpurun_out/r4u; mkdir -p $O
( timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -k "gradw or reference_order" </dev/null ) > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
timeout 60 python tools/bspmm_probe.py </dev/null 2>&1 | grep -v amdgpu | tee $O/bspmm_probe.txt
