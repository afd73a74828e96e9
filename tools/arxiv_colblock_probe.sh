#!/bin/bash
B="--workload arxiv --steps 50 --warmup 5 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for cb in 64 0; do for ex in 1 0; do
  GGL_COL_BLOCK=$cb GGL_EXACT_LONG_ROWS=$ex timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('arxiv col_block=$cb exact=$ex ms/step', round(d['ms_per_step'],4), 'aggregate', round(d['roofline']['ms_per_aggregate'],4), 'launches', d['roofline']['launches_per_aggregate'])"
done; done
