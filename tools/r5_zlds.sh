#!/bin/bash
O=gpurun_out/${1:-r5z}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q -k "gat or GAT or headmean or head_mean" ) > $O/pytest_gat_zlds.log 2>&1; tail -3 $O/pytest_gat_zlds.log
timeout 600 python tools/r5_gat_probe.py 2>&1 | grep -v amdgpu | tee $O/r5_gat_probe.txt
( timeout 600 python bench.py --workload reddit-gat --steps 5 --warmup 2 --secondary off --pmc-traffic off ) 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reddit-gat bench ms_per_step', d['ms_per_step'], 'parity', d['parity'])"
