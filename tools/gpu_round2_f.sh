#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1"
run sage    timeout 600 bash -c "python tools/sage_minibatch_probe.py > $O/sage_minibatch.txt 2>&1"
run local   timeout 600 bash -c "python tools/locality_probe.py > $O/locality.txt 2>&1"
run example timeout 600 bash -c "python examples/sage_trainer_amd.py --n_epoch 1 --nodes 100000 > $O/example_sage.txt 2>&1"
run bench   timeout 900 bash -c "python bench.py --no-cpu-baseline > $O/bench_products_nocpu.json 2> $O/bench_products.err"
run rocprofG timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_g && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o r2 -- python $R/tools/gat_lastlayer_probe.py > $R/$O/rocprof_gat_model.log 2>&1; cp \$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1) $R/$O/r2_gat_model_kernel_stats.csv"
cat $O/summary.txt; tail -6 $O/pytest_gpu.log | cut -c1-300
grep -v "amdgpu.ids" $O/sage_minibatch.txt | tail -9; grep -v amdgpu $O/locality.txt; tail -5 $O/example_sage.txt
cut -c1-330 $O/bench_products_nocpu.json; echo
python tools/prof_summary.py $O/r2_gat_model_kernel_stats.csv 16
