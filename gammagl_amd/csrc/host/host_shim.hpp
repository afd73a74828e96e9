// gammagl_amd/csrc/host/host_shim.hpp — minimal host stand-in for the HIP device environment: the HOST BUILD of the
// kernel sources (-DGGL_EMULATE), i.e. the CPU backend of the ops.
//
// The reference's ops dispatch on x.is_cuda() / x.is_cpu() (gammagl/mpops/torch_ext/src/segment_sum.cpp:19-33): CPU
// tensors run its serial C++ loops, and BASELINE config 1 (Cora, `--gpu -1`) is exactly that path.  The row kernels of
// reduce.hip / plan.hip / backward.hip / gat.hip / epilogue.hip / sample.hip use no LDS, no cross-lane shuffles, no
// barriers and no order-dependent atomics (the sampler's first-occurrence atomicMin is order-independent), so running
// their bodies one thread at a time on the host executes the same arithmetic in the same order — on rows reduced in
// one piece bit for bit what the GPU computes and what the reference's serial loops compute.  Kernels that DO use LDS
// or shuffles (gat_fast.hip, hub16.hip, the wide-head GAT backward, the LDS path of edgedot.hip) are GPU-build only;
// gpu_only_stubs.cpp answers their capability queries with "no" and the generic kernels serve those shapes.
//
// Two consumers: `make -C gammagl_amd/csrc host` -> gammagl_amd/lib/libggl_mpops_host.so, bound to the `CPU` dispatch
// key (gammagl_amd/torch_ops.py; CPU tensors ONLY — a GPU tensor never reaches it, and a missing HIP library still
// fails loudly), and tests/emul/build.sh -> the same sources at -O1 / under AddressSanitizer for the `-m "not gpu"`
// suite's kernel-logic tests.  Single-threaded on purpose: the reference's shipped CPU extension is serial too
// (its OpenMP macro is never defined, setup.py:50).  Never the oracle: oracle/ is a separate restatement.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

struct ggl_emul_dim3 { unsigned x = 1, y = 1, z = 1; };
inline thread_local ggl_emul_dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

typedef void *hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDeviceToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToDevice };
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
  std::memset(d, v, n);
  return hipSuccess;
}

static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
template <typename T> static inline T __builtin_amdgcn_readfirstlane(T v) { return v; }

// order-independent atomics only (min): one host thread at a time, so a plain update is the same result
static inline long long atomicMin(long long *p, long long v) { const long long o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int *p, int v) { const int o = *p; if (v > o) *p = v; return o; }

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct double2 { double x, y; };
struct uint4 { uint32_t x, y, z, w; };

namespace ggl_emul {
template <typename F> static inline void launch(int64_t grid, int64_t block, F &&body) {
  gridDim.x = (unsigned)grid;
  blockDim.x = (unsigned)block;
  for (int64_t b = 0; b < grid; ++b) {
    blockIdx.x = (unsigned)b;
    for (int64_t t = 0; t < block; ++t) {
      threadIdx.x = (unsigned)t;
      body();
    }
  }
}
template <typename F> static inline void launch2d(unsigned gx, unsigned gy, int64_t block, F &&body) {
  gridDim.x = gx;
  gridDim.y = gy;
  blockDim.x = (unsigned)block;
  for (unsigned by = 0; by < gy; ++by) {
    blockIdx.y = by;
    for (unsigned bx = 0; bx < gx; ++bx) {
      blockIdx.x = bx;
      for (int64_t t = 0; t < block; ++t) {
        threadIdx.x = (unsigned)t;
        body();
      }
    }
  }
  blockIdx.y = 0;
  gridDim.y = 1;
}
}  // namespace ggl_emul
