#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
( timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "random_vs_oracle or ragged or strided or all_dtypes or special_values or int_vector or fuzz or cpp_registered or folded" </dev/null ) > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
timeout 60 python tools/ragged_probe.py </dev/null 2>&1 | grep -v amdgpu | tee $O/ragged_probe.txt
