#!/bin/bash
# Round-2 artifact refresh at HEAD: suites, smoke, bench (+ degree order, arxiv), rocprof of the bench command,
# secondary configs, mini-batch probe, self-halo overhead, the papers100M-sized 8-way share.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 1500 bash -c "python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1"
run smoke    timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run bench    timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run benchdeg timeout 900 bash -c "python bench.py --relabel degree --no-cpu-baseline --pmc-traffic off > $O/bench_products_degree.json 2> $O/bench_products_degree.err"
run bencharx timeout 900 bash -c "python bench.py --workload arxiv --steps 100 --warmup 10 --no-cpu-baseline --pmc-traffic off > $O/bench_arxiv.json 2> $O/bench_arxiv.err"
run rocprofB timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $R/$O/rocprof_bench.json 2> $R/$O/rocprof_bench.err; cp \$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $R/$O/bench_kernel_stats.csv"
run cfg      timeout 1200 bash -c "python tools/bench_configs.py 2 3 4 > $O/configs_2_3_4.txt 2>&1"
run sagemb   timeout 900 bash -c "python tools/sage_minibatch_probe.py > $O/sage_minibatch.txt 2>&1"
run gatll    timeout 600 bash -c "python tools/gat_lastlayer_probe.py > $O/gat_last_layer.txt 2>&1"
run halo     timeout 900 bash -c "python tools/self_halo_probe.py products > $O/self_halo.txt 2>&1"
run share    timeout 1500 bash -c "python tools/share_probe.py papers100M 8 3 $O/share_papers.json > $O/share_papers.txt 2>&1"
cat $O/summary.txt; tail -3 $O/pytest_gpu.log | cut -c1-200; tail -1 $O/smoke.log
python tools/prof_summary.py $O/bench_kernel_stats.csv 16 > $O/bench_summary.txt; head -8 $O/bench_summary.txt | cut -c1-150
python -c "
import json
for f in ('$O/bench_products.json','$O/bench_products_degree.json','$O/bench_arxiv.json','$O/rocprof_bench.json'):
    d=json.load(open(f)); tf=d['config']['transform_first']; print(f, round(d['ms_per_step'],2), round(d['value']/1e9,3), d['config']['aggregations_per_step'], (round(tf['ms_per_step'],2), round(tf['value']/1e9,3)) if tf else None, d['roofline']['ms_per_launch'], round(d['roofline']['frac'],3), d['roofline']['traffic'])
d=json.load(open('$O/bench_products.json')); print(d.get('cpu_baseline',{}).get('value'))
"
