#!/bin/bash
# what the driver runs at round end (pytest -m gpu, smoke, the default bench) + the artefacts profiles/ cites
O=gpurun_out/${1:-r4final}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"
timeout 600 python tools/rccl_capture_retry.py > $O/rccl_capture_retry.txt 2>&1; cat $O/rccl_capture_retry.txt
( python tools/op_overhead_probe.py; python tools/op_overhead_probe.py --nodes 169343 --edges 2315598 --width 256 --reps 200 ) > $O/op_call_overhead.txt 2>&1
timeout 600 python tools/ops_shape_sweep.py products > $O/ops_shape_sweep_products.txt 2>&1
bash tools/gpu_r4_profiles.sh ${1:-r4final}/prof > $O/profiles.log 2>&1; tail -5 $O/profiles.log
