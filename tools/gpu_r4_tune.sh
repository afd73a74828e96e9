#!/bin/bash
# the measured halo-chunk count (DistGCNTrainer.tune_halo_chunks) on dry products-sized shares + the exact_long_max cap
O=gpurun_out/r4p; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_order or long_row or integration or hip_ext" ) > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
B="--no-cpu-baseline --pmc-traffic off --no-comparison --secondary off"
for P in 8 4; do for tune in 1 0; do
  GGL_HALO_TUNE=$tune timeout 400 python bench.py --workload products --dry-parts $P --steps 20 --warmup 5 $B 2>$O/dry${P}_tune${tune}.err | tee $O/dry${P}_tune${tune}.json | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); x=d.get('exchange') or d.get('config',{}).get('exchange') or {}
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
    return None
print('dry P=$P tune=$tune ms/step', round(d['ms_per_step'],3), 'halo_chunks', find(d,'halo_chunks'))"
done; done
# products single GPU twice (noise check of the headline after the ABI 7 rebuild)
for i in 1 2; do timeout 300 python bench.py --workload products --steps 20 --warmup 5 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('products run $i ms/step', round(d['ms_per_step'],4), 'aggregate', round(d['roofline']['ms_per_aggregate'],4))"; done
