#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "long_row or reference_order" ) > $O/parity_long.log 2>&1
tail -3 $O/parity_long.log
B="python bench.py --steps 10 --warmup 3 --no-comparison --no-cpu-baseline --pmc-traffic off --secondary off"
for mode in "1 1" "1 0" "0 1"; do
  set -- $mode
  GGL_EXACT_LONG_ROWS=$1 GGL_EXACT_SIDE_STREAM=$2 timeout 600 $B > $O/bench_exact$1_side$2.json 2> $O/bench_exact$1_side$2.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_exact$1_side$2.json") if l.startswith("{")][-1])
print("exact=$1 side=$2 ms_per_step", round(d["ms_per_step"],3), "ms_per_aggregate", round(d["roofline"]["ms_per_aggregate"],3))
PY
done
for mode in "1" "0"; do
  GGL_EXACT_LONG_ROWS=$mode timeout 600 $B --workload arxiv --steps 30 > $O/arxiv_exact$mode.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("$O/arxiv_exact$mode.json") if l.startswith("{")][-1])
print("arxiv exact=$mode ms_per_step", round(d["ms_per_step"],4), "ms_per_aggregate", round(d["roofline"]["ms_per_aggregate"],4))
PY
done
cd /tmp && rm -rf /tmp/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o t -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err
python $R/tools/trace_timeline.py $(find /tmp/prof_b -name '*kernel_trace.csv' | head -1) multi_tensor_apply 0.05 > $R/$O/timeline.txt 2>&1
head -14 $R/$O/timeline.txt | cut -c1-130
