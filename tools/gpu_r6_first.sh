#!/bin/bash
# round 6, first GPU call: the new at-size parity tests (printed errors), the max-backward footprint probe
O=gpurun_out/${1:-r6a}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
free -g | head -2; nproc
( time timeout 1500 python -m pytest tests/test_gpu_refsize.py -m gpu -q -s -k "headmean or gat_model or mean_backward or 16bit" ) > $O/pytest_refsize_new.log 2>&1
grep -E "err vs fp64|passed|failed|Error|assert" $O/pytest_refsize_new.log | head -60
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layers or golden or max_backward" ) > $O/pytest_parity_new.log 2>&1
tail -4 $O/pytest_parity_new.log
timeout 900 python tools/r6_probe.py maxbwd > $O/r6_maxbwd_probe.txt 2>&1; cat $O/r6_maxbwd_probe.txt
