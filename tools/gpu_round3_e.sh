#!/bin/bash
# Round 3, fifth GPU pass: suites at HEAD; bspmm weight gradient (Q = 12 slabs for 44-channel heads); 16-bit sums with
# shifted ragged tails + 16-byte hub producers + short consumer chains; per-graph XCD runs (auto) A/B; planted bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3e; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 900 bash -c "python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1"
run bspmm    timeout 600 bash -c "python tools/bspmm_bwd_probe.py $O/bspmm_bwd.txt > $O/bspmm.log 2>&1"
run half     timeout 600 bash -c "python tools/half_probe.py $O/half.txt > $O/half.log 2>&1"
run roword   timeout 900 bash -c "python tools/roworder_probe.py $O/roworder.txt > $O/roworder.log 2>&1"
run planted  timeout 900 bash -c "python bench.py --workload products-planted --pmc-traffic l2 --no-cpu-baseline > $O/bench_planted.json 2> $O/bench_planted.err"
run plantedc timeout 900 bash -c "python bench.py --workload products-planted --relabel cluster --also-relabel none --pmc-traffic l2 --no-cpu-baseline > $O/bench_planted_cluster.json 2> $O/bench_planted_cluster.err"
cat $O/summary.txt; tail -8 $O/pytest_gpu.log | cut -c1-220
cat $O/bspmm_bwd.txt $O/half.txt $O/roworder.txt 2>/dev/null | cut -c1-330
tail -n 2 $O/bspmm.log $O/half.log $O/roworder.log | cut -c1-300
python - <<'PY'
import json
for f in ("bench_planted","bench_planted_cluster"):
    try:
        d=json.load(open("gpurun_out/r3e/"+f+".json"))
    except Exception as ex:
        print(f, "no json", ex); continue
    c=d["config"]; rf=d["roofline"]
    print("==",f, round(d["ms_per_step"],2), round(d["value"]/1e9,3), "AF:", (c.get("aggregate_first") or {}).get("ms_per_step"))
    print("   orderings", c.get("orderings"))
    print("   roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in rf.items() if k not in ("kernel","achieved_basis","traffic_source")})
PY
