#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1"
run smoke   timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run pmcgat  timeout 900 bash tools/pmc_gat.sh
mkdir -p profiles; cp $O/pmc_gat_reddit.json profiles/r2_pmc_gat_reddit.json 2>/dev/null
run cfg3    timeout 900 bash -c "python tools/bench_configs.py 3 > $O/configs_3.txt 2>&1"
run cfg24   timeout 900 bash -c "python tools/bench_configs.py 2 4 > $O/configs_2_4.txt 2>&1"
run bench   timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run benchdeg timeout 900 bash -c "python bench.py --relabel degree --no-cpu-baseline > $O/bench_products_degree.json 2> $O/bench_products_degree.err"
run rocprofB timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/rocprof_bench.json 2> $R/$O/rocprof_bench.err; cp \$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $R/$O/r2_bench_kernel_stats.csv"
run torchrun1 timeout 600 bash -c "python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload arxiv --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err"
cat $O/summary.txt; tail -6 $O/pytest_gpu.log | cut -c1-300; tail -1 $O/smoke.log
grep -v amdgpu $O/configs_3.txt | cut -c1-700; grep -v amdgpu $O/configs_2_4.txt | cut -c1-300
cut -c1-330 $O/bench_products.json; echo; cut -c1-330 $O/bench_products_degree.json; echo; cut -c1-200 $O/bench_torchrun1.json; echo
python tools/prof_summary.py $O/r2_bench_kernel_stats.csv 12
