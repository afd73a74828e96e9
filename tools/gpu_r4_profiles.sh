#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats summaries of every BASELINE config's bench command + a dry 8-way products share
O=gpurun_out/${1:-r4j}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null

prof() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python $R/bench.py "$@" --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_$name.json 2> $R/$O/bench_$name.err )
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f $O/${name}_kernel_stats.csv; python tools/prof_summary.py $O/${name}_kernel_stats.csv 16 > $O/${name}_summary.txt; fi
  t=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
  if [ -n "$t" ]; then python tools/trace_timeline.py $t multi_tensor_apply 0.05 > $O/${name}_timeline.txt 2>&1; fi
  echo "== $name"; head -8 $O/${name}_summary.txt | cut -c1-150
}
prof products --steps 5 --warmup 2
prof arxiv --workload arxiv --steps 20 --warmup 5
prof reddit_gat --workload reddit-gat --steps 4 --warmup 2
prof sage_minibatch --workload sage-minibatch --steps 40 --warmup 10
prof products_dry8 --dry-parts 8 --steps 5 --warmup 2
prof products_dry4 --dry-parts 4 --steps 5 --warmup 2
prof papers_share --workload papers-share --steps 3 --warmup 1
