#!/bin/bash
# round 6: a graph WITH locality (hierarchical planted communities, products-sized) in three node orders — ms per aggregate,
# fabric-side traffic over compulsory, L2 hit rate each (round-5 verdict item 8)
O=gpurun_out/${1:-r6p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for R in random cluster none; do
  ( time timeout 1200 python bench.py --workload products-planted --relabel $R --also-relabel none --steps 10 --warmup 3 \
      --secondary off --pmc-traffic l2 --no-cpu-baseline ) > $O/planted_$R.out 2> $O/planted_$R.err
  echo "== relabel=$R rc=$?"; tail -c 2500 $O/planted_$R.out; tail -3 $O/planted_$R.err
  cp bench_detail.json $O/planted_${R}_detail.json 2>/dev/null
done
