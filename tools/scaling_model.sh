#!/bin/bash
# products-sized graph: ONE rank's share of a P-way partition (dry: send lists and buffers as in the real run, nothing
# on the wire) timed on one MI355X through bench.py itself, P = 2, 4, 8 -> gpurun_out/scaling_model.jsonl;
# tools/scaling_model.py turns it into the predicted N-GPU bench lines (compute measured, link time modelled).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/scaling_model.jsonl
for P in 2 4 8; do
  timeout 600 python bench.py --dry-parts $P --no-cpu-baseline --pmc-traffic off --also-relabel none > $O/share_products_$P.json 2> $O/share_products_$P.err
  python -c "
import json; d=json.load(open('$O/share_products_$P.json')); c=d['config']
print(json.dumps({'parts': $P, 'train_step_ms': d['ms_per_step'], 'aggregate_first_ms': (c.get('aggregate_first') or {}).get('ms_per_step'),
  'halo_rows': c['rank0_halo_rows'], 'send_rows': c['rank0_send_rows'], 'owned_rows': c['rank0_owned_rows'], 'local_edges': c['rank0_local_edges'],
  'local_source_edges': c['rank0_local_source_edges'], 'halo_floats_per_row': c['halo_floats_per_row_per_step'], 'a2a_GB_in': c['exchange']['a2a_GB_in']}))" >> $O/scaling_model.jsonl
done
cat $O/scaling_model.jsonl
python tools/scaling_model.py $O/scaling_model.jsonl > $O/scaling_model.txt; cat $O/scaling_model.txt
