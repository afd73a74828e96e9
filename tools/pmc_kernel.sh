#!/bin/bash
# HBM-side traffic of the SpMM-sum kernel at width K on the bench graph: separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md §HBM) -> gpurun_out/pmc_<wl>_k<K>.json.   usage: tools/pmc_kernel.sh products 256
WL=${1:-products}; K=${2:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/pmc_probe.py $WL $K > $O/pmc_${WL}_k${K}_$tag.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
L=$(find /tmp/pmc_TCC_HIT_sum -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summary.py $O/pmc_${WL}_k${K}.json $F $W $L | grep row_reduce | cut -c1-400
grep "^E " $O/pmc_${WL}_k${K}_FETCH_SIZE.log
