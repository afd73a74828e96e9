#!/bin/bash
# round 6: what the driver runs at round end — the whole -m gpu suite, smoke(), the default bench (its command line)
O=gpurun_out/${1:-r6full}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$?"; tail -n 1 $O/bench_default.out | head -c 4200; echo; tail -4 $O/bench_default.err
cp bench_detail.json $O/ 2>/dev/null
