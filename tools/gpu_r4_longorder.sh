#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_order or long_row or int_vector or cpp_registered or integration or hip_ext" ) > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
B="--no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for w in arxiv products; do for i in 1 2; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$w run $i ms/step', round(d['ms_per_step'],4), 'aggregate', round(d['roofline']['ms_per_aggregate'],4))"
done; done
timeout 300 python tools/narrow16_probe.py 2>&1 | grep -v amdgpu | tee $O/narrow16_after.txt | head -4
