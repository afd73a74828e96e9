#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
( timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "gradw or reference_order" </dev/null ) > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
for pr in 0 1; do GGL_EXACT_SIDE_PRIORITY=$pr timeout 150 python tools/bspmm_probe.py </dev/null 2>&1 | grep -v amdgpu | tee -a $O/bspmm_probe.txt; done
B="--no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for pr in 0 1; do GGL_EXACT_SIDE_PRIORITY=$pr timeout 120 python bench.py --workload products --steps 20 --warmup 5 $B </dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('products prio=$pr ms/step', round(d['ms_per_step'],4), 'aggregate', round(d['roofline']['ms_per_aggregate'],4))" | tee -a $O/bspmm_probe.txt; done
