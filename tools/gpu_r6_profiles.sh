#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats summaries of every BASELINE config's bench command + dry 8- / 4-way products shares,
# and the FETCH / WRITE / L2 counters of the GAT kernels (incl. the output layer's gat_sh_*)
O=gpurun_out/${1:-r6prof}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
prof() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python $R/bench.py "$@" --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_$name.json 2> $R/$O/bench_$name.err )
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f $O/${name}_kernel_stats.csv; python tools/prof_summary.py $O/${name}_kernel_stats.csv 16 > $O/${name}_kernel_summary.txt; fi
  t=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
  if [ -n "$t" ]; then python tools/trace_timeline.py $t multi_tensor_apply ${MINMS:-0.05} > $O/${name}_timeline.txt 2>&1; fi
  echo "== $name"; head -8 $O/${name}_kernel_summary.txt | cut -c1-150
}
prof products --steps 5 --warmup 2
prof arxiv --workload arxiv --steps 20 --warmup 5
prof reddit_gat --workload reddit-gat --steps 4 --warmup 2
MINMS=0.0 prof sage_minibatch --workload sage-minibatch --steps 40 --warmup 10
prof products_dry8 --dry-parts 8 --steps 5 --warmup 2
prof products_dry4 --dry-parts 4 --steps 5 --warmup 2
prof papers_share --workload papers-share --steps 3 --warmup 1
bash tools/pmc_gat.sh > $O/pmc_gat.log 2>&1; grep "gat_sh\|gat_fwd2\|gat_bwd" $O/pmc_gat.log | cut -c1-260; cp gpurun_out/pmc_gat_reddit.json $O/pmc_gat_reddit.json 2>/dev/null
