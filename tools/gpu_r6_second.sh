#!/bin/bash
# round 6, second GPU call: packed pair dots of the head-mean GAT backward (A/B + parity), the fixed refsize tests
O=gpurun_out/${1:-r6b}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests/test_gpu_refsize.py -m gpu -q -s -k "headmean or gat_model" ) > $O/pytest_refsize_gat.log 2>&1
grep -E "err vs fp64|passed|failed|Error|assert" $O/pytest_refsize_gat.log | head -40
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q -s -k "headmean or fusedgat or max_backward" ) > $O/pytest_gat_forms.log 2>&1
grep -E "full-size head-mean|passed|failed|Error|assert" $O/pytest_gat_forms.log | head -30
timeout 900 python tools/r6_probe.py gat > $O/r6_gat_pk_probe.txt 2>&1; cat $O/r6_gat_pk_probe.txt
