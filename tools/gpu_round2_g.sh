#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1"
run smoke   timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run bench   timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run benchdeg timeout 900 bash -c "python bench.py --relabel degree --no-cpu-baseline --pmc-traffic off > $O/bench_products_degree.json 2> $O/bench_products_degree.err"
run rocprofB timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $R/$O/rocprof_bench.json 2> $R/$O/rocprof_bench.err; cp \$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $R/$O/r2_bench_kernel_stats.csv"
run halo    timeout 900 bash -c "python tools/self_halo_probe.py products > $O/self_halo.txt 2>&1"
run share   timeout 1500 bash -c "python tools/share_probe.py papers100M 8 3 $O/share_papers.json > $O/share_papers.txt 2>&1"
run cfg2    timeout 900 bash -c "python tools/bench_configs.py 2 > $O/configs_2.txt 2>&1"
cat $O/summary.txt; tail -3 $O/pytest_gpu.log | cut -c1-200; tail -1 $O/smoke.log
python -c "
import json
for f in ('$O/bench_products.json','$O/bench_products_degree.json','$O/rocprof_bench.json'):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],2), round(d['value']/1e9,3), d['config']['aggregations_per_step'], d['config']['transform_first'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['roofline']['traffic'])
d=json.load(open('$O/bench_products.json')); print(d.get('cpu_baseline',{}).get('value'))
print(open('$O/share_papers.json').read())
"
grep -v "amdgpu\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|socket" $O/self_halo.txt | tail -4; grep -v amdgpu $O/configs_2.txt | head -3
python tools/prof_summary.py $O/r2_bench_kernel_stats.csv 14
