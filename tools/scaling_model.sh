#!/bin/bash
# products-sized graph: ONE rank's share of a P-way partition (dry: send lists and buffers as in the real run,
# nothing on the wire) timed on one MI355X, P = 1, 2, 4, 8 -> gpurun_out/scaling_model.jsonl.
# tools/scaling_model.py turns it into the predicted N-GPU bench lines (compute measured, link time modelled).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/scaling_model.jsonl
for P in 2 4 8; do
  r=$((P / 2))
  timeout 600 python tools/share_probe.py products $P $r $O/share_products_$P.json > $O/share_products_$P.txt 2>&1
  python -c "import json; d=json.load(open('$O/share_products_$P.json')); d['parts']=$P; print(json.dumps(d))" >> $O/scaling_model.jsonl
done
cat $O/scaling_model.jsonl | cut -c1-400
