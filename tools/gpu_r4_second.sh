#!/bin/bash
# round 4, second GPU call: the exact-order hub rows (hubf32.hip) — parity first, then what they cost
mkdir -p gpurun_out/r4b
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "long_row or reference_order or folded or strided or epilogue" ) > gpurun_out/r4b/parity_long.log 2>&1
tail -4 gpurun_out/r4b/parity_long.log
( time timeout 900 python -m pytest tests/test_gpu_refsize.py -x -q ) > gpurun_out/r4b/refsize.log 2>&1
tail -4 gpurun_out/r4b/refsize.log
B="python bench.py --steps 10 --warmup 3 --no-comparison --no-cpu-baseline --pmc-traffic off --secondary off"
for mode in "1 1" "1 0" "0 1"; do
  set -- $mode
  GGL_EXACT_LONG_ROWS=$1 GGL_EXACT_SIDE_STREAM=$2 timeout 600 $B > gpurun_out/r4b/bench_exact$1_side$2.json 2> gpurun_out/r4b/bench_exact$1_side$2.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4b/bench_exact$1_side$2.json") if l.startswith("{")][-1])
print("exact=$1 side=$2 ms_per_step", round(d["ms_per_step"],3), "ms_per_aggregate", round(d["roofline"]["ms_per_aggregate"],3))
PY
done
for mode in "1 1" "0 1"; do
  set -- $mode
  GGL_EXACT_LONG_ROWS=$1 timeout 600 $B --workload arxiv --steps 30 > gpurun_out/r4b/arxiv_exact$1.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4b/arxiv_exact$1.json") if l.startswith("{")][-1])
print("arxiv exact=$1 ms_per_step", round(d["ms_per_step"],4), "ms_per_aggregate", round(d["roofline"]["ms_per_aggregate"],4))
PY
done
