#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
( timeout 40 python -m pytest tests/test_gpu_parity.py -x -q -k "gspmm_bspmm_golden or random_vs_oracle or cpp_registered" </dev/null ) > $O/pytest_subset.log 2>&1; tail -1 $O/pytest_subset.log
timeout 25 python tools/waves_probe.py arg32 </dev/null 2>&1 | grep -v amdgpu | grep "max" | tee $O/max_bwd.txt
