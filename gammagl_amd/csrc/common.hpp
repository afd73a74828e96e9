// gammagl_amd/csrc/common.hpp — shared host/device helpers for libggl_mpops_hip.so (gfx950 only).
#pragma once
#ifdef GGL_EMULATE
// Host build of the SAME kernel sources, one "thread" at a time (csrc/host/host_shim.hpp): the CPU backend of the
// ops (libggl_mpops_host.so, CPU dispatch key) and the engine of the GPU-less container's kernel-logic tests.
#include "host/host_shim.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include <cfloat>
#include <climits>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/ggl_mpops.h"

namespace ggl {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kBlock = 256;      // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;

enum Op { OP_SUM = 0, OP_MEAN = 1, OP_MAX = 2 };

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define GGL_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::ggl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                       __LINE__);                                                        \
      return GGL_EHIP;                                                                   \
    }                                                                                    \
  } while (0)
#define GGL_REQUIRE(cond, code, ...)                                                     \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      ::ggl::set_error(__VA_ARGS__);                                                     \
      return (code);                                                                     \
    }                                                                                    \
  } while (0)
#define GGL_LAUNCH_CHECK() GGL_HIP_CHECK(hipGetLastError())

// kernel launch: KERN is a parenthesised kernel name, e.g. (k<float, 4>).  A dispatch carries its grid as
// 32-bit WORK-ITEM counts per dimension, so gridDim.x * blockDim.x must stay below 2^32: a 256-thread
// block per 4 rows overflows that at 67 M rows (found on the papers100M-sized graph: the tail of the rows
// was silently not launched).  Grids wider than max_grid_x blocks are therefore folded into (x, y) and
// kernels that index by block use block_id(); every such kernel already returns for ids past its work.
int64_t max_grid_x();
static inline void fold_grid(int64_t grid, unsigned *gx, unsigned *gy) {
  const int64_t mx = max_grid_x();
  if (grid <= mx) { *gx = (unsigned)(grid > 0 ? grid : 1); *gy = 1; return; }
  *gx = (unsigned)mx;
  *gy = (unsigned)((grid + mx - 1) / mx);
}
#ifdef GGL_EMULATE
#define GGL_LAUNCH(KERN, GRID, BLOCK, STREAM, ...)                                       \
  do {                                                                                   \
    unsigned ggl_gx_, ggl_gy_;                                                           \
    ::ggl::fold_grid((GRID), &ggl_gx_, &ggl_gy_);                                        \
    ::ggl_emul::launch2d(ggl_gx_, ggl_gy_, (BLOCK), [&]() { KERN(__VA_ARGS__); });       \
  } while (0)
#else
#define GGL_LAUNCH(KERN, GRID, BLOCK, STREAM, ...)                                       \
  do {                                                                                   \
    unsigned ggl_gx_, ggl_gy_;                                                           \
    ::ggl::fold_grid((GRID), &ggl_gx_, &ggl_gy_);                                        \
    hipLaunchKernelGGL(KERN, dim3(ggl_gx_, ggl_gy_), dim3((unsigned)(BLOCK)), 0, (STREAM), __VA_ARGS__); \
  } while (0)
#endif
// linear block index of a (possibly folded) launch
__device__ __forceinline__ int64_t block_id() { return (int64_t)blockIdx.y * (int64_t)gridDim.x + (int64_t)blockIdx.x; }
__device__ __forceinline__ int64_t thread_id() { return block_id() * (int64_t)blockDim.x + (int64_t)threadIdx.x; }
__device__ __forceinline__ int64_t grid_threads() { return (int64_t)gridDim.x * (int64_t)gridDim.y * (int64_t)blockDim.x; }

struct Options {
  int64_t unroll = 4;        // neighbour loads in flight per lane in the f32 fast path (4 or 8)
  int64_t unroll_narrow = 16; // ... and where a row owns <= 4 lanes (K <= 16 floats): 4 or 16
  int64_t unroll_narrow_max = 0;  // ... for max too (A/B knob: lost with 64-bit argmax registers in round 1; re-measured in round 4)
  // 1 = give each XCD a contiguous range of row blocks (private-L2 locality).  OFF by default: measured
  // on MI355X (profiles/kbench_r1.txt) it changes nothing on a randomly ordered graph and is 4x SLOWER
  // on a degree-ordered one (one XCD inherits all the hub rows); round-robin is the load balancer.
  int64_t xcd_swizzle = 0;
  int64_t force_generic = 0; // route f32 through the VEC=1 generic kernel (A/B aid)
  int64_t col_block = 64;    // wide f32 SpMM-sum / mean: launches over column blocks of this width (0 = one launch)
  int64_t col_block_min_degree = 24;      // ... and only where a row averages at least this many edges (reuse to find)
  int64_t col_block_min_edges = 8000000;  // below this many edges the blocks are twice as wide (launch-bound graphs)
  int64_t ragged4 = 1;       // f32 rows that are not aligned float4s (K % 4 != 0): 4 floats per lane + ragged last lane
  int64_t ragged_max = 1;    // ... for segment_max as well (0 = the one-element-per-lane kernels of rounds 2-3: an A/B knob)
  // 0 = natural row order; 1 = length-sorted rows where several rows share a wavefront (balances
  // the lanes of a wave); 2 = also for the wave-per-row kernels (heavy rows first)
  int64_t row_order = 1;
  int64_t max_grid_x = 1 << 22;  // blocks per grid row before a launch is folded into 2-D (tests lower it)
  // f32 sums: rows longer than the plan's chunk are added up in the reference's serial order (hubf32.hip: bit-identical to
  // the CPU extension on EVERY row) instead of chunk by chunk (within rounding of it); 0 = the chunked walk (A/B switch)
  int64_t exact_long_rows = 1;
  // hosts: gspmm max backward (products-sized graph, forward + backward in ms, profiles/r5_max_backward.txt):
  //             int64 witnesses   int32 witnesses   winner mask: forward order (writelane / select)   scattered
  //   K =  64        16.8              12.7                        14.0 / 14.4                           15.4
  //   K = 128        32.4              25.0                        21.6 / 23.1                           22.2
  //   K = 256        67.1              53.0                        41.1 / 44.4                           41.7
  int64_t maxbwd_arg32 = 1;        // witnesses from a compact int32 copy (ggl_spmm_max_bwd32) ...
  int64_t maxbwd_mask = 128;       // ... and from this many columns up a 1-bit winner mask instead (0 = never)
  int64_t maxbwd_mask_kmax = 256;  // ... up to this many columns (the mask is an E x K/8-byte transient; K = 602 measured slower AND 12 GiB on the
                                   //     Reddit-sized graph: ggl_policy_maxbwd_form; 0 = no upper bound: tests force the mask at any width)
  int64_t maxbwd_mask_wlane = 1;   // ... its forward-order records assembled with v_writelane (inline asm; 0 = selects)
  int64_t maxbwd_mask_cols = 0;    // ... its walk in 64-column blocks like the plain sum's (A/B: loses, the record is re-read per block)
  int64_t maxbwd_mask_scatter = 0; // ... its records scattered to transposed positions instead of kept in forward order (A/B)
  int64_t exact_long_max = (int64_t)1 << 21;   // exact_long_rows: unless the plan's longest row is longer than this (0 = no limit)
  int64_t exact_side_stream = 1;  // ... launched beside the walk over the other rows (0 = in front of it, same stream)
  // the head-mean (output-layer) GAT walks, round 5 (Reddit-sized graph, profiles/r5_gat_sh_forms.txt: layer fwd 5.13 -> 4.64 ms,
  // fwd + bwd 20.9 -> 18.6 ms, 2-layer step 32.0 -> 29.6 ms):
  int64_t gat_sh_pk = 1;          // backward walks of the head-mean GAT: dots packed over head pairs (v_pk_fma_f32), select-free reduce-scatter (round 6)
  int64_t gat_sh_pipe = 0;        // source walk of the head-mean backward with its gathers software-pipelined one step ahead (A/B, round 6)
  int64_t gat_sh_glds = 0;        // destination walk of the backward: the row's G in per-lane LDS slots too (A/B)
  int64_t gat_sh_prefetch = 1;    // forward / destination walks request the next step's ids before this step's gathers
  int64_t gat_sh_zlds = 1;        // source walk of the backward: the row's z_j in per-lane LDS slots instead of 32 registers (140 -> 125:
                                  // 4 wavefronts per SIMD without spills) + the same id prefetch
  int64_t gat_sh_waves = 0;       // >= 4: the output-layer GAT backward's source walk (dropout form) built for 4 wavefronts per SIMD (A/B)
  // static-shape sampler hop: count / flag + scan (+ clamp) as ONE launch each (single-pass chained scan).  Measured (round 5,
  // profiles/r5_sage_fused_scans.txt): 76 -> 63 launches per replayed step, but the same 0.17 ms for the two hops — the fused
  // kernels take the 13-33 us their look-back chains need where five 5-us launches stood.  OFF: no gain to set against a
  // kernel that spins on its predecessors (an A/B knob only: its look-back assumes lower-numbered blocks are resident or
  // will be scheduled, and its slots are told apart by a tag of (seed, offset, scan, block) instead of being zeroed per hop).
  int64_t hop_fused_scans = 0;
  // small hops (round 6, A/B): the first scan + clamp + clamp_last as ONE single-workgroup launch (<= 65 536 rows: both hops of the reference's
  // [25, 10] mini-batch), the flag scan likewise while it covers <= 32 768 positions.  Measured in three forms (values computed inside the
  // workgroup: step 0.62 -> 0.76 ms; thread-contiguous scan: +0.055 ms; coalesced wave-slice scan: +0.02 .. +0.1 ms against the rocprim
  // launches in paired runs) — a single workgroup on one CU is not faster than four 5-us launches it replaces.  OFF.
  int64_t hop_small_scans = 0;
  // the hub walk's side queue created with the device's greatest priority (big eager launches only: hubf32.hip).  OFF: the
  // isolated aggregate gains 1 % (13.75 -> 13.60 ms) but the products STEP nothing (75.46 vs 75.45 ms) and a partitioned step
  // LOSES (dry 8-way share 14.3 -> 16.9 ms, 4-way 26.1 -> 29.3: profiles/r5_priority_ab.txt)
  int64_t hub_priority = 0;
  int64_t hub_pipe = 1;           // hub walk's consumer with its LDS reads software-pipelined (0 = round 4's: A/B, heavy configuration only)
  int64_t hub_one_launch = 2;     // hub walk once per aggregate over the full width: 1 = always, 0 = once per column block, 2 = where the long rows lead the ids
};
Options &options();

// Philox4x32-10 keyed on (seed, offset); one 4-word draw per vector of 4 outputs (epilogue.hip, reduce.hip)
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint64_t index, uint64_t offset, uint64_t seed) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

int rng_advance(int64_t *rng_state, void *stream);  // offset += 1 on the stream (epilogue.hip)

#ifndef GGL_EMULATE
// hubf32.hip: the long rows of an f32 sum in serial order, one partial row each (see the file's header)
struct HubF32Args {
  const float *x;
  int64_t x_ld;
  const int32_t *perm;      // segment mode: element of sorted position p; SpMM: weight index of p when !w_by_pos
  const int32_t *col;       // SpMM: source row of sorted position p (NULL = segment mode)
  const float *w;           // edge weights or NULL
  int w_by_pos;
  int64_t H, C;             // C > 0: multi-head weights w[wi * H + column / C]
  const int64_t *rowptr;
  const int32_t *long_rows;
  const int32_t *long_order;   // positions in long_rows, longest row first (or NULL)
  int64_t n_long;
  int64_t K;                // columns of this launch (a column block of a wider matrix: x points at its first column)
  float *partial;           // [n_long, K]
  int64_t avg_long_len;     // average length of the long rows (picks the stage size)
  int f64;                  // segment sums of doubles: x / x_ld / K / partial in 4-byte WORDS (2 per element), see hubf32.hip
};
int hub_f32_launch(const HubF32Args &a, hipStream_t stream, bool beside, int *forked);   // *forked: 0 or the join token
int hub_f32_join(hipStream_t stream, int token);
#endif

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- dtype semantics: "accumulate in the storage dtype" (segment_sum_cpu.cpp:56) ---------------
// S = storage type in memory, A = register type.  add() rounds to storage precision after every
// step, exactly like c10::Half / c10::BFloat16 operator+= (float add, then round-to-nearest-even).
struct bf16_t { uint16_t bits; };
struct f16_t { uint16_t bits; };

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t x = __float_as_uint(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;  // c10::BFloat16: NaN -> 0x7FC0
  return (uint16_t)((x + (((x >> 16) & 1u) + 0x7fffu)) >> 16);
}
__device__ __forceinline__ float f16_to_f32(uint16_t b) {
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
  _Float16 h = (_Float16)f;  // v_cvt_f16_f32, round-to-nearest-even
  uint16_t b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}

template <typename T> struct TT;

#define GGL_INT_TT(T, LOW)                                                               \
  template <> struct TT<T> {                                                             \
    using S = T;                                                                         \
    using A = T;                                                                         \
    static __device__ __forceinline__ A load(S v) { return v; }                          \
    static __device__ __forceinline__ S store(A v) { return v; }                         \
    static __device__ __forceinline__ A add(A a, A b) {                                  \
      using U = typename std::make_unsigned<T>::type;                                    \
      return (T)(U)((U)a + (U)b);                                                        \
    }                                                                                    \
    static __device__ __forceinline__ bool less(A a, A b) { return a < b; }              \
    static __device__ __forceinline__ A lowest() { return (T)(LOW); }                    \
    static __device__ __forceinline__ A zero() { return (T)0; }                          \
    static __device__ __forceinline__ A count(int64_t c) { return (T)c; }                \
    static __device__ __forceinline__ bool gt1(A c) { return c > (T)1; }                 \
    static __device__ __forceinline__ A div(A a, A c) { return (T)(a / c); }             \
  };
GGL_INT_TT(uint8_t, 0)
GGL_INT_TT(int8_t, INT8_MIN)
GGL_INT_TT(int16_t, INT16_MIN)
GGL_INT_TT(int32_t, INT32_MIN)
GGL_INT_TT(int64_t, INT64_MIN)

template <> struct TT<float> {
  using S = float;
  using A = float;
  static __device__ __forceinline__ A load(S v) { return v; }
  static __device__ __forceinline__ S store(A v) { return v; }
  static __device__ __forceinline__ A add(A a, A b) { return __fadd_rn(a, b); }
  static __device__ __forceinline__ bool less(A a, A b) { return a < b; }
  static __device__ __forceinline__ A lowest() { return -FLT_MAX; }
  static __device__ __forceinline__ A zero() { return 0.0f; }
  // the reference counts in x's dtype with += 1 (segment_mean_cpu.cpp:44,52): a float counter
  // stops growing at 2^24
  static __device__ __forceinline__ A count(int64_t c) { return (float)(c < 16777216 ? c : 16777216); }
  static __device__ __forceinline__ bool gt1(A c) { return c > 1.0f; }
  static __device__ __forceinline__ A div(A a, A c) { return __fdiv_rn(a, c); }
};
template <> struct TT<double> {
  using S = double;
  using A = double;
  static __device__ __forceinline__ A load(S v) { return v; }
  static __device__ __forceinline__ S store(A v) { return v; }
  static __device__ __forceinline__ A add(A a, A b) { return __dadd_rn(a, b); }
  static __device__ __forceinline__ bool less(A a, A b) { return a < b; }
  static __device__ __forceinline__ A lowest() { return -DBL_MAX; }
  static __device__ __forceinline__ A zero() { return 0.0; }
  static __device__ __forceinline__ A count(int64_t c) { return (double)c; }
  static __device__ __forceinline__ bool gt1(A c) { return c > 1.0; }
  static __device__ __forceinline__ A div(A a, A c) { return __ddiv_rn(a, c); }
};
template <> struct TT<f16_t> {
  using S = uint16_t;
  using A = float;  // always holds a value exactly representable in f16
  static __device__ __forceinline__ A load(S v) { return f16_to_f32(v); }
  static __device__ __forceinline__ S store(A v) { return f32_to_f16(v); }
  static __device__ __forceinline__ A add(A a, A b) { return f16_to_f32(f32_to_f16(__fadd_rn(a, b))); }
  static __device__ __forceinline__ bool less(A a, A b) { return a < b; }
  static __device__ __forceinline__ A lowest() { return -65504.0f; }
  static __device__ __forceinline__ A zero() { return 0.0f; }
  static __device__ __forceinline__ A count(int64_t c) { return (float)(c < 2048 ? c : 2048); }
  static __device__ __forceinline__ bool gt1(A c) { return c > 1.0f; }
  static __device__ __forceinline__ A div(A a, A c) { return f16_to_f32(f32_to_f16(__fdiv_rn(a, c))); }
};
template <> struct TT<bf16_t> {
  using S = uint16_t;
  using A = float;  // always holds a value exactly representable in bf16
  static __device__ __forceinline__ A load(S v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ S store(A v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ A add(A a, A b) { return bf16_to_f32(f32_to_bf16(__fadd_rn(a, b))); }
  static __device__ __forceinline__ bool less(A a, A b) { return a < b; }
  static __device__ __forceinline__ A lowest() { return bf16_to_f32(0xFF7Fu); }
  static __device__ __forceinline__ A zero() { return 0.0f; }
  static __device__ __forceinline__ A count(int64_t c) { return (float)(c < 256 ? c : 256); }
  static __device__ __forceinline__ bool gt1(A c) { return c > 1.0f; }
  static __device__ __forceinline__ A div(A a, A c) { return bf16_to_f32(f32_to_bf16(__fdiv_rn(a, c))); }
};

// eight 16-bit elements at a 2-byte aligned address as ONE 16-byte access (the backend emits global_load_dwordx4 for the
// packed struct: unaligned access mode) — rows of f16 / bf16 whose width is not a multiple of 8 (reduce.hip, hub16.hip)
struct __attribute__((packed, aligned(2))) H8U { uint16_t v[8]; };

static inline size_t dtype_size(int dtype) {
  switch (dtype) {
    case GGL_U8: case GGL_I8: return 1;
    case GGL_I16: case GGL_F16: case GGL_BF16: return 2;
    case GGL_I32: case GGL_F32: return 4;
    case GGL_I64: case GGL_F64: return 8;
    default: return 0;
  }
}

// block id -> logical block id so that the blocks one XCD executes (block b runs on XCD b % 8, observed) cover RUNS of
// consecutive row blocks: neighbouring rows of a locality-ordered graph then share one private 4 MiB L2 instead of
// being dealt round-robin to all eight.  swizzle = 1: one contiguous eighth of the launch per XCD (4x slower on a
// degree-ordered graph: one XCD inherits every heavy row); swizzle = B >= 2: runs of B consecutive blocks, the eight
// XCDs working on eight adjacent runs (balanced whatever the order).  Pure performance: any bijection is correct.
// (guide §5.5 T1)
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nb, int swizzle) {
  if (!swizzle) return b;
  if (swizzle == 1) {
    const int64_t per = nb >> 3;
    const int64_t main_blocks = per << 3;
    if (b >= main_blocks) return b;  // ragged tail keeps identity
    return (b & 7) * per + (b >> 3);
  }
  const int64_t B = swizzle;
  const int64_t main_blocks = (nb / (8 * B)) * (8 * B);
  if (b >= main_blocks) return b;
  const int64_t xcd = b & 7, k = b >> 3;          // the k-th block this XCD executes
  return ((k / B) * 8 + xcd) * B + (k % B);
}

}  // namespace ggl
