#!/bin/bash
# round 5, first GPU call: the new paths' parity tests, the A/B probe, the default bench (the driver's command)
O=gpurun_out/${1:-r5a}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_rows or max_backward or strided or epilogue" ) > $O/pytest_new.log 2>&1
tail -4 $O/pytest_new.log
( time timeout 900 python -m pytest tests/test_gpu_refsize.py -m gpu -x -q ) > $O/pytest_refsize.log 2>&1
tail -4 $O/pytest_refsize.log
timeout 900 python tools/r5_probe.py hub max > $O/r5_probe.txt 2>&1; cat $O/r5_probe.txt
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$?"; tail -c 4500 $O/bench_default.out; tail -5 $O/bench_default.err
cp bench_detail.json $O/ 2>/dev/null
