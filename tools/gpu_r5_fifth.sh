#!/bin/bash
# round 5, fifth GPU call: fused sampler scans + overflow fold, hub defaults (per-block + priority queue); full suite + default bench + sage timeline
O=gpurun_out/${1:-r5j}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$?"; tail -n 1 $O/bench_default.out | cut -c1-3800; tail -4 $O/bench_default.err
cp bench_detail.json $O/bench_default_detail.json 2>/dev/null
for FS in 1 0; do
  rm -rf /tmp/prof_sage$FS
  ( cd /tmp && GGL_HOP_FUSED_SCANS=$FS timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sage$FS -o t -- python $R/bench.py --workload sage-minibatch --steps 40 --warmup 10 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_sage_fs$FS.json 2> $R/$O/bench_sage_fs$FS.err )
  t=$(find /tmp/prof_sage$FS -name '*kernel_trace.csv' | head -1)
  python tools/trace_timeline.py $t multi_tensor_apply 0.0 > $O/sage_fs${FS}_timeline.txt 2>&1
  head -1 $O/sage_fs${FS}_timeline.txt; tail -n 1 $O/bench_sage_fs$FS.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_scans=$FS ms_per_step', d['ms_per_step'])"
done
