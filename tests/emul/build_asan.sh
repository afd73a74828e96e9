#!/bin/bash
# AddressSanitizer build of the host-emulated kernels (see csrc/host/host_shim.hpp).
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
SRC=../../gammagl_amd/csrc
OUT=libggl_emul_asan.so
if [ -f $OUT ] && [ -z "$(find $SRC ../../include -newer $OUT -type f)" ]; then exit 0; fi
$CXX -DGGL_EMULATE -x c++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -shared \
  -ffp-contract=off -Wno-unused-function $SRC/plan.hip $SRC/reduce.hip $SRC/backward.hip $SRC/edgedot.hip $SRC/gat.hip $SRC/epilogue.hip $SRC/sample.hip $SRC/host/gpu_only_stubs.cpp -o $OUT
