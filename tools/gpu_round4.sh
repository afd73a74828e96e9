#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; rm -rf $O/prof_* ; mkdir -p $O; : > $O/summary.txt
R=$PWD
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1200 bash -c "python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1"
run products timeout 900 bash -c "python bench.py > $O/bench_products.json 2> $O/bench_products.err"
run rocprof timeout 900 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_bench.json 2> $R/$O/rocprof.err"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do n=$(echo $c | cut -d' ' -f1)
run pmc_$n timeout 600 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/prof_pmc_$n -o p -- python $R/tools/pmc_probe.py > $R/$O/pmc_$n.log 2>&1"
done
find $O -name "*.db" -delete; find $O -type f -size +20M -delete
python tools/pmc_summary.py $O/pmc_spmm_k256.json $O/prof_pmc_FETCH_SIZE/p_counter_collection.csv $O/prof_pmc_WRITE_SIZE/p_counter_collection.csv $O/prof_pmc_TCC_HIT_sum/p_counter_collection.csv | cut -c1-300
cat $O/summary.txt; tail -4 $O/pytest_gpu.log; cut -c1-1500 $O/bench_products.json; tail -3 $O/bench_products.err
python tools/prof_summary.py $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) 24
tail -1 $O/pmc_FETCH_SIZE.log
