#!/bin/bash
# tests/emul/build.sh — host build of the kernel sources (one thread at a time) for logic tests in
# the GPU-less container.  Same sources as the product's CPU backend (csrc/host/host_shim.hpp), built at -O1 for the tests.
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
command -v "$CXX" >/dev/null 2>&1 || CXX=g++
SRC=../../gammagl_amd/csrc
OUT=libggl_emul.so
if [ -f $OUT ] && [ -z "$(find $SRC ../../include -newer $OUT -type f)" ]; then exit 0; fi
$CXX -DGGL_EMULATE -x c++ -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-function \
  $SRC/plan.hip $SRC/reduce.hip $SRC/backward.hip $SRC/edgedot.hip $SRC/gat.hip $SRC/epilogue.hip $SRC/sample.hip $SRC/host/gpu_only_stubs.cpp -o $OUT
