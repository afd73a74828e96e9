#!/bin/bash
# one gpurun call: GPU suite, bench, rocprofv3 kernel stats, PMC passes (FETCH_SIZE / WRITE_SIZE apart)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; rm -rf $O/prof_* ; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1200 bash -c "python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1"
run products timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run rocprof timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_bench.json 2> $R/$O/rocprof.err"
run pmc_fetch timeout 600 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/prof_pmc_fetch -o f -- python $R/tools/pmc_probe.py > $R/$O/pmc_fetch.log 2>&1"
run pmc_write timeout 600 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/prof_pmc_write -o w -- python $R/tools/pmc_probe.py > $R/$O/pmc_write.log 2>&1"
run pmc_l2 timeout 600 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$O/prof_pmc_l2 -o l -- python $R/tools/pmc_probe.py > $R/$O/pmc_l2.log 2>&1"
# keep only small CSVs
find $O -name "*.db" -delete; find $O -type f -size +20M -delete
du -sh $O; find $O -type f | head -40
cat $O/summary.txt; tail -12 $O/pytest_gpu.log; cat $O/bench_products.json; tail -3 $O/bench_products.err
S=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); echo "== $S"; head -25 "$S"
for d in fetch write l2; do C=$(find $O/prof_pmc_$d -name "*counter_collection.csv" | head -1); echo "== $C"; grep -i "row_reduce" "$C" | head -8; done
tail -2 $O/pmc_fetch.log
