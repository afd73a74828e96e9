#!/bin/bash
# per-rank step time of dry P-way products shares against the long-row threshold (GGL_LONG_ROW; 0 = the automatic rule)
B="--steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for P in 8 4; do
 for c in 0 2048 1024 512 256 128; do
  GGL_LONG_ROW=$c timeout 300 python bench.py --dry-parts $P $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('P=$P chunk=$c ms/step', round(d['ms_per_step'],3), 'halo aggregate', round(d['roofline']['ms_per_aggregate'],3))"
 done
done
for c in 0 2048 1024; do
  GGL_LONG_ROW=$c timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('P=1 chunk=$c ms/step', round(d['ms_per_step'],3), 'aggregate', round(d['roofline']['ms_per_aggregate'],3))"
done
