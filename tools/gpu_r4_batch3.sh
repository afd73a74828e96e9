#!/bin/bash
O=gpurun_out/${1:-r4k}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -m gpu -x -q -s -k "not refsize" ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log; grep "eager all_reduce" $O/pytest_gpu.log
timeout 600 python tools/rccl_capture_retry.py > $O/rccl_capture_retry.txt 2>&1; cat $O/rccl_capture_retry.txt
B="--no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for P in 8 4 2; do
  timeout 600 python bench.py --dry-parts $P --steps 10 --warmup 3 $B > $O/bench_dry$P.json 2> $O/bench_dry$P.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_dry$P.json") if l.startswith("{")][-1])
print("dry parts $P ms_per_step", round(d["ms_per_step"],3))
PY
done
timeout 600 python tools/ops_shape_sweep.py products > $O/ops_shape_sweep_products.txt 2>&1; grep -E "K=  7|K= 16|C=  7" $O/ops_shape_sweep_products.txt | head -20
timeout 600 python bench.py --workload sage-minibatch --steps 200 --warmup 20 $B > $O/bench_sage.json 2>/dev/null; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_sage.json") if l.startswith("{")][-1]); print("sage ms/step", d["ms_per_step"], d["value"])
PY
