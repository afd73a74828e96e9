#!/bin/bash
B="python bench.py --workload sage-minibatch --steps 300 --warmup 30 --no-cpu-baseline --pmc-traffic off --secondary off"
for i in 1 2 3; do
  for e in 1 0; do
    GGL_EXACT_LONG_ROWS=$e timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('exact=$e run $i ms/step', round(d['ms_per_step'],4))"
  done
done
