#!/bin/bash
# round 5: what the driver runs at round end (pytest -m gpu, smoke, the default bench), then the profiles of record
O=gpurun_out/${1:-r5final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$?"; tail -n 1 $O/bench_default.out | cut -c1-3800; tail -4 $O/bench_default.err
cp bench_detail.json $O/bench_default_detail.json 2>/dev/null
bash tools/gpu_r5_profiles.sh ${1:-r5final}/prof > $O/profiles.log 2>&1; tail -40 $O/profiles.log
timeout 600 python tools/soak2_probe.py > $O/soak.txt 2>&1; tail -6 $O/soak.txt
