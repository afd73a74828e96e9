#!/bin/bash
# HBM-side traffic of the fused GAT kernels on the Reddit-sized graph -> gpurun_out/pmc_gat_reddit.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmcg_$tag
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcg_$tag -o p -- python $R/tools/pmc_gat_probe.py > $O/pmc_gat_$tag.log 2>&1
done
F=$(find /tmp/pmcg_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/pmcg_WRITE_SIZE -name "*counter_collection.csv" | head -1)
L=$(find /tmp/pmcg_TCC_HIT_sum -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summary.py $O/pmc_gat_reddit.json $F $W $L | grep "gat_" | cut -c1-330
