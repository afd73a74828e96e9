#!/bin/bash
O=gpurun_out/${1:-r4i}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 600 python tools/ops_shape_sweep.py products > $O/ops_shape_sweep_products.txt 2>&1; tail -3 $O/ops_shape_sweep_products.txt
( python tools/op_overhead_probe.py; python tools/op_overhead_probe.py --nodes 169343 --edges 2315598 --width 256 --reps 200 ) > $O/op_call_overhead.txt 2>&1; tail -12 $O/op_call_overhead.txt
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; tail -c 300 $O/bench_default.err
