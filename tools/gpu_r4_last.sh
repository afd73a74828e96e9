#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
( timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged or all_dtypes or special_values" </dev/null ) > $O/pytest_subset.log 2>&1; tail -1 $O/pytest_subset.log
timeout 40 python tools/f16max_probe.py </dev/null 2>&1 | grep -v amdgpu | tee $O/f16max_probe.txt
