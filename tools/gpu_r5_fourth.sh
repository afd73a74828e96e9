#!/bin/bash
# round 5, fourth GPU call: four double sums in the fast GAT destination walk; the masked max backward in one launch
O=gpurun_out/${1:-r5d}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "max_backward or gat or cpp_registered or fused" ) > $O/pytest_new.log 2>&1
tail -4 $O/pytest_new.log
( time timeout 900 python -m pytest tests/test_gpu_refsize.py -m gpu -x -q -s ) > $O/pytest_refsize.log 2>&1
tail -4 $O/pytest_refsize.log; grep "err vs fp64" $O/pytest_refsize.log
timeout 900 python tools/r5_probe.py max > $O/r5_probe_max.txt 2>&1; cat $O/r5_probe_max.txt
( timeout 600 python bench.py --workload reddit-gat --steps 5 --warmup 2 --secondary off ) > $O/bench_gat.out 2> $O/bench_gat.err
tail -n 1 $O/bench_gat.out | cut -c1-2600; cp bench_detail.json $O/bench_gat_detail.json 2>/dev/null
timeout 900 python tools/ops_shape_sweep.py products > $O/ops_shape_sweep_products.txt 2>&1; tail -5 $O/ops_shape_sweep_products.txt
