#!/bin/bash
# round 6: the products profile again (timed plan with its row order), and what the first layer aggregated before its transform
# does to one rank's dry share (the option --aggregate-first; NOT the association the headline keeps)
O=gpurun_out/${1:-r6prof2}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
name=products
rm -rf /tmp/prof_$name
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off > $R/$O/bench_$name.json 2> $R/$O/bench_$name.err )
f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
cp $f $O/${name}_kernel_stats.csv; python tools/prof_summary.py $O/${name}_kernel_stats.csv 16 > $O/${name}_kernel_summary.txt
t=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $t multi_tensor_apply 0.05 > $O/${name}_timeline.txt 2>&1
head -6 $O/${name}_kernel_summary.txt | cut -c1-150; tail -n1 $O/bench_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['ms_per_aggregate'], d['roofline']['ms_per_launch'])"
for P in 8 4; do
  for AF in "" "--aggregate-first"; do
    timeout 600 python bench.py --dry-parts $P --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off $AF > $O/dry${P}${AF}.json 2> $O/dry${P}${AF}.err
    tail -n1 $O/dry${P}${AF}.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dry P=$P $AF', round(d['ms_per_step'],2), 'ms/step', d['config'].get('association'), d['config'].get('aggregations_per_step'))"
  done
done
