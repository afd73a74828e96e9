#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
for cfg in "GGL_UNROLL=4" "GGL_UNROLL=8" "GGL_UNROLL=8 GGL_ROW_ORDER=2" "GGL_UNROLL=4 GGL_ROW_ORDER=2" "GGL_UNROLL=8 GGL_LONG_ROW=128" "GGL_UNROLL=4 GGL_LONG_ROW=128" "GGL_UNROLL=4 GGL_LONG_ROW=512"; do
  echo "== $cfg"; env $cfg python tools/narrow_probe.py arxiv 2>&1 | grep -v amdgpu | grep "K=  64\|K= 256\|K=  16" | cut -c1-230
done > $O/unroll_probe.txt 2>&1
echo "== products GGL_UNROLL=8" >> $O/unroll_probe.txt; GGL_UNROLL=8 python tools/narrow_probe.py products 2>&1 | grep "K=  64\|K= 256\|K=  16" | cut -c1-230 >> $O/unroll_probe.txt
cat $O/unroll_probe.txt
