#!/bin/bash
# refresh the dry-share rocprof summaries after the halo-chunk tuner (subset of gpu_r4_profiles.sh; every read guarded)
O=gpurun_out/r4t; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
prof() {
  name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python $R/bench.py "$@" --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off </dev/null > $R/$O/bench_$name.json 2> $R/$O/bench_$name.err )
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f $O/${name}_kernel_stats.csv; python tools/prof_summary.py $O/${name}_kernel_stats.csv 16 > $O/${name}_summary.txt </dev/null; fi
  t=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
  if [ -n "$t" ]; then python tools/trace_timeline.py $t multi_tensor_apply 0.05 > $O/${name}_timeline.txt 2>&1 </dev/null; fi
  echo "== $name"; [ -f $O/${name}_summary.txt ] && head -6 $O/${name}_summary.txt | cut -c1-150
}
prof products_dry8 --dry-parts 8 --steps 5 --warmup 2
prof products_dry4 --dry-parts 4 --steps 5 --warmup 2
