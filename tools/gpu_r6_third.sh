#!/bin/bash
# round 6, third GPU call: per-kernel durations of the head-mean GAT walks, round-5 form vs packed pair dots, and where the
# source walk's wave cycles go (SQ counters)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6c}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for PK in 0 1; do
  rm -rf /tmp/kt_$PK
  GGL_GAT_SH_PK=$PK timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$PK -o p -- python $R/tools/pmc_gat_probe.py > $O/kt_pk$PK.log 2>&1
  F=$(find /tmp/kt_$PK -name "*kernel_stats.csv" | head -1)
  echo "== GGL_GAT_SH_PK=$PK"; python $R/tools/prof_summary.py $F 14 | tee $O/gat_kernels_pk$PK.txt | grep -E "gat_|total"
done
for PK in 0 1; do
  rm -rf /tmp/sq_$PK
  GGL_GAT_SH_PK=$PK timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/sq_$PK -o p -- python $R/tools/pmc_gat_probe.py > $O/sq_pk$PK.log 2>&1
  F=$(find /tmp/sq_$PK -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY' | tee $O/gat_sq_pk$PK.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    if "gat_sh" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    n = max(cnt[(k, c)] for c in d)
    print(k, {c: round(v / n / 1e6, 2) for c, v in d.items()}, "(millions per launch)")
PY
done
