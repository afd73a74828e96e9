#!/bin/bash
# Round-3 artifact refresh at HEAD: suites, smoke, the default bench line, rocprofv3 --kernel-trace --stats of the bench
# command (+ the step's timeline), every other bench workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final3; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 1200 bash -c "python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1"
run smoke    timeout 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
run bench    timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run rocprofB timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off --also-relabel none > $R/$O/rocprof_bench.json 2> $R/$O/rocprof_bench.err; cp \$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $R/$O/bench_kernel_stats.csv; python $R/tools/trace_timeline.py \$(find /tmp/prof_b -name '*kernel_trace.csv' | head -1) > $R/$O/bench_timeline.txt 2>&1"
run planted  timeout 1200 bash -c "python bench.py --workload products-planted --pmc-traffic l2 --no-cpu-baseline > $O/bench_products_planted.json 2> $O/bench_products_planted.err"
run share    timeout 900 bash -c "python bench.py --workload papers-share --steps 3 --warmup 1 > $O/bench_papers_share.json 2> $O/bench_papers_share.err"
run gat      timeout 600 bash -c "python bench.py --workload reddit-gat > $O/bench_reddit_gat.json 2> $O/bench_reddit_gat.err"
run sage     timeout 600 bash -c "python bench.py --workload sage-minibatch --steps 200 --warmup 20 > $O/bench_sage_minibatch.json 2> $O/bench_sage_minibatch.err"
run arxiv    timeout 600 bash -c "python bench.py --workload arxiv --steps 100 --warmup 10 > $O/bench_arxiv.json 2> $O/bench_arxiv.err"
cat $O/summary.txt; tail -12 $O/pytest_gpu.log | cut -c1-220; tail -1 $O/smoke.log
python tools/prof_summary.py $O/bench_kernel_stats.csv 18 > $O/bench_summary.txt; head -10 $O/bench_summary.txt | cut -c1-160
head -40 $O/bench_timeline.txt | cut -c1-160
python - <<'PY'
import json
for f in ("bench_products","rocprof_bench","bench_products_planted","bench_papers_share","bench_reddit_gat","bench_sage_minibatch","bench_arxiv"):
    try:
        d=json.load(open("gpurun_out/final3/"+f+".json"))
    except Exception as ex:
        print(f, "no json", ex); continue
    c=d["config"]; rf=d["roofline"] or {}
    print("==",f, round(d["ms_per_step"],3), round(d["value"]/1e9,3), "AF:", (c.get("aggregate_first") or {}).get("ms_per_step"))
    for o in c.get("orderings") or []:
        print("    order", {k:(round(v,4) if isinstance(v,float) else v) for k,v in o.items() if k!="traffic_source"})
    print("   roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in rf.items() if k not in ("kernel","achieved_basis","traffic_source")})
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], (d["cpu_baseline"].get("torch_fallback") or {}).get("value"))
PY
