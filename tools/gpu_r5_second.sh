#!/bin/bash
# round 5, second GPU call: the whole -m gpu suite + smoke + the default bench (what the driver runs), GAT A/B + PMC, dry shares
O=gpurun_out/${1:-r5b}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
grep "err vs fp64" $O/pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$?"; tail -n 1 $O/bench_default.out | cut -c1-3600; tail -4 $O/bench_default.err
cp bench_detail.json $O/ 2>/dev/null
timeout 600 python tools/r5_gat_probe.py > $O/r5_gat_probe.txt 2>&1; cat $O/r5_gat_probe.txt
bash tools/pmc_gat.sh > $O/pmc_gat.log 2>&1; tail -12 $O/pmc_gat.log; cp gpurun_out/pmc_gat_reddit.json $O/ 2>/dev/null
for P in 8 4; do
  ( timeout 600 python bench.py --dry-parts $P --steps 10 --warmup 3 --no-comparison --no-cpu-baseline --pmc-traffic off --secondary off ) > $O/bench_dry$P.out 2> $O/bench_dry$P.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_dry$P.out").read().strip().splitlines()[-1])
    print("dry share P=$P: ms_per_step", d["ms_per_step"], "value", d["value"])
except Exception as ex:
    print("dry $P failed", ex)
PY
done
