#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; : > $O/summary.txt
run() { name=$1; shift; local t0=$(date +%s); "$@"; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $O/summary.txt; }
run pytest  timeout 1500 bash -c "python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1"
run sage    timeout 600 bash -c "python tools/sage_minibatch_probe.py > $O/sage_minibatch.txt 2>&1"
run narrowA timeout 600 bash -c "python tools/narrow_probe.py arxiv > $O/narrow_arxiv.txt 2>&1"
run narrowP timeout 600 bash -c "python tools/narrow_probe.py products > $O/narrow_products.txt 2>&1"
run halost  timeout 900 bash -c "python tools/halo_stats.py products > $O/halo_stats.txt 2>&1"
run share   timeout 1500 bash -c "python tools/share_probe.py papers100M 8 3 $O/share_papers.json > $O/share_papers.txt 2>&1"
run pmc16   timeout 900 bash tools/pmc_kernel.sh products 16
run pmc32   timeout 900 bash tools/pmc_kernel.sh products 32
run pmc256  timeout 900 bash tools/pmc_kernel.sh products 256
cat $O/summary.txt; tail -15 $O/pytest_gpu.log | cut -c1-300
grep -v "amdgpu.ids" $O/sage_minibatch.txt; grep -v "amdgpu.ids" $O/narrow_arxiv.txt; grep -v "amdgpu.ids" $O/narrow_products.txt
grep -v "amdgpu.ids" $O/halo_stats.txt; tail -4 $O/share_papers.txt | cut -c1-1500
for k in 16 32 256; do python -c "
import json; d=json.load(open('$O/pmc_products_k$k.json'))
for n,e in d.items():
    if 'row_reduce' in n: print('K=$k', n[:70], {a:(round(b['avg']) if isinstance(b,dict) else b) for a,b in e.items()})
"; done
