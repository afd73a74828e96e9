#!/bin/bash
B="--steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for c in 0 2048 8192 16384; do
  GGL_LONG_ROW=$c timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('P=1 chunk=$c ms/step', round(d['ms_per_step'],3), 'aggregate', round(d['roofline']['ms_per_aggregate'],3))"
done
for P in 8 4; do for ex in 0 1; do
  GGL_DIST_EXACT=$ex timeout 300 python bench.py --dry-parts $P $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('dry P=$P dist_exact=$ex ms/step', round(d['ms_per_step'],3))"
done; done
