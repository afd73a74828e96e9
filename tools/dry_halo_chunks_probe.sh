#!/bin/bash
B="--steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for P in 8 4; do
 for c in 4 2 1; do
  GGL_HALO_CHUNKS=$c timeout 300 python bench.py --dry-parts $P $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('P=$P halo column chunks=$c ms/step', round(d['ms_per_step'],3))"
 done
done
