#!/bin/bash
# the whole GPU suite + the default bench command (what the driver runs), artefacts under gpurun_out/<tag>
O=gpurun_out/${1:-r4h}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; tail -c 400 $O/bench_default.err
