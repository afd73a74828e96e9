#!/bin/bash
# round 4, first GPU call: the at-size reference parity tests, the default bench line (parity + secondary + calibration),
# the list of DRAM / EA counters this part exposes
mkdir -p gpurun_out/r4a
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 900 python -m pytest tests/test_gpu_refsize.py -x -q ) > gpurun_out/r4a/refsize.log 2>&1
tail -5 gpurun_out/r4a/refsize.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r4a/bench_default.err
rocprofv3 -L 2>/dev/null | grep -i -E "dram|EA0|EA_|MALL|HBM|FETCH|WRITE_SIZE" | head -80 > gpurun_out/r4a/counters.txt
wc -l gpurun_out/r4a/counters.txt
