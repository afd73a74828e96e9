#!/bin/bash
# Round 3, sixth GPU pass: suites at HEAD; bspmm (pipelined tail); half (f32 tail reverted); planted graph in the
# generator's own order (what perfect clustering would give: L2 hit rate / traffic ceiling).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3f; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest   timeout 900 bash -c "python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1"
run bspmm    timeout 600 bash -c "python tools/bspmm_bwd_probe.py $O/bspmm_bwd.txt > $O/bspmm.log 2>&1"
run half     timeout 600 bash -c "python tools/half_probe.py $O/half.txt > $O/half.log 2>&1"
run plantedo timeout 900 bash -c "python bench.py --workload products-planted --relabel none --also-relabel none --pmc-traffic l2 --no-cpu-baseline --no-comparison > $O/bench_planted_oracle.json 2> $O/bench_planted_oracle.err"
cat $O/summary.txt; tail -4 $O/pytest_gpu.log | cut -c1-220
cat $O/bspmm_bwd.txt $O/half.txt 2>/dev/null | cut -c1-330
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3f/bench_planted_oracle.json")); rf=d["roofline"]
print("planted, generator's order:", round(d["ms_per_step"],2), {k:(round(v,4) if isinstance(v,float) else v) for k,v in rf.items() if k not in ("kernel","achieved_basis","traffic_source")})
PY
