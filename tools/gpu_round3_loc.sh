#!/bin/bash
# locality artefacts at HEAD (cluster_order with many labels): planted bench (3 node orders, L2 counters), halo statistics
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final3; mkdir -p $O
timeout 1200 python bench.py --workload products-planted --pmc-traffic l2 --no-cpu-baseline > $O/bench_products_planted.json 2> $O/bench_products_planted.err; echo "planted rc=$?"
timeout 900 python tools/halo_stats.py products > $O/halo_stats.txt 2>&1; echo "halo rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/final3/bench_products_planted.json")); c=d["config"]; rf=d["roofline"]
print(round(d["ms_per_step"],2), "AF", c["aggregate_first"]["ms_per_step"])
for o in c["orderings"]: print({k:(round(v,4) if isinstance(v,float) else v) for k,v in o.items() if k!="traffic_source"})
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in rf.items() if k in ("traffic_per_aggregate","l2_hit_rate","frac","ms_per_aggregate","traffic_over_compulsory")})
PY
grep -v amdgpu $O/halo_stats.txt | cut -c1-200
