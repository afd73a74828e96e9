#!/bin/bash
# Round 3, third GPU pass: suites at HEAD; bspmm weight gradient A/B (scratch-free kernel); 16-bit segment sums A/B;
# bench lines: products (default), products-planted with L2 counters, papers-share; op sweep at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 900 bash -c "python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1"
run bspmm    timeout 600 bash -c "python tools/bspmm_bwd_probe.py $O/bspmm_bwd.txt > $O/bspmm.log 2>&1"
run half     timeout 600 bash -c "python tools/half_probe.py $O/half.txt > $O/half.log 2>&1"
run bench    timeout 600 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run planted  timeout 900 bash -c "python bench.py --workload products-planted --pmc-traffic l2 --no-cpu-baseline > $O/bench_planted.json 2> $O/bench_planted.err"
run share    timeout 900 bash -c "python bench.py --workload papers-share --steps 3 --warmup 1 > $O/bench_papers_share.json 2> $O/bench_papers_share.err"
run sweep    timeout 900 bash -c "python tools/ops_shape_sweep.py products > $O/ops_shape_sweep_products.txt 2>&1"
cat $O/summary.txt; tail -8 $O/pytest_gpu.log | cut -c1-220
cat $O/bspmm_bwd.txt $O/half.txt 2>/dev/null; tail -2 $O/bspmm.log $O/half.log | cut -c1-300
python - <<'PY'
import json
for f in ("bench_products","bench_planted","bench_papers_share"):
    try:
        d=json.load(open("gpurun_out/r3c/"+f+".json"))
    except Exception as ex:
        print(f, "no json", ex); continue
    c=d["config"]; rf=d["roofline"]
    print("==",f, round(d["ms_per_step"],2), round(d["value"]/1e9,3), "AF:", (c.get("aggregate_first") or {}).get("ms_per_step"))
    print("   orderings", c.get("orderings"))
    print("   roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in rf.items() if k not in ("kernel","achieved_basis")})
    print("   exchange", c.get("exchange"))
PY
tail -c 600 $O/bench_papers_share.err
cat $O/ops_shape_sweep_products.txt | tail -45
