#!/bin/bash
# Offline GEMM selection for bench.py: PyTorch TunableOp times every rocBLAS / hipBLASLt solution for each f32 GEMM shape
# of the step (both trainers of the bench run) and writes the winners to gammagl_amd/tuned/tunableop_gfx950_<workload>.csv,
# which bench.py loads with tuning OFF.  ~2 GPU-minutes.   usage: tools/tune_gemms.sh [products|arxiv|sage-minibatch|reddit-gat]
WL=${1:-products}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out gammagl_amd/tuned
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_$WL.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100 PYTORCH_TUNABLEOP_VERBOSE=0
# (--hipgraph off: tuning times each solution with host syncs, which a stream capture does not allow)
timeout 1500 python bench.py --workload $WL --no-cpu-baseline --pmc-traffic off --no-tuned-gemm --hipgraph off --warmup 4 > gpurun_out/tune_$WL.json 2> gpurun_out/tune_$WL.err
F=gammagl_amd/tuned/tunableop_gfx950_${WL//-/_}.csv
cp gpurun_out/tunableop_${WL}0.csv $F
cat $F | cut -c1-160
