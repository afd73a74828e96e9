// gammagl_amd/csrc/gat_common.hpp — what gat.hip (every head shape, also built for the host-emulated tests) and
// gat_fast.hip (the GPU-only fast paths) share: launch dimensions, the attention-dropout word, small helpers, and
// the declarations of the hub-chunk combine kernels defined in gat.hip.
#pragma once
#include "common.hpp"

#ifndef GGL_EMULATE
#define GGL_EXPF(x) expf(x)
#else
#define GGL_EXPF(x) std::exp(x)
#endif

namespace ggl {

struct GatDims {
  float slope;
  int64_t N, H, C, K, E;
  int64_t chunk, n_long, n_chunks, chunk_blocks, nblocks;
  uint32_t drop_thresh;  // attention dropout (gat_conv.py:104, GATConvFuse's last argument): keep when
  float drop_scale;      // Philox(p * H + h).x >= drop_thresh, kept alphas scaled by 1 / (1 - p)
  int64_t es;  // element stride of alpha / de: 1 = two [E,H] arrays, 2 = interleaved [E,H,2] (one 64-byte line per edge)
  int logL;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : __fmul_rn(v, slope); }

__device__ __forceinline__ uint32_t pick_word(const U4 &u, int c) {
  return c == 0 ? u.x : (c == 1 ? u.y : (c == 2 ? u.z : u.w));
}
// Attention-dropout word of (sorted position p, head h): counter-based, ~15 VALU instructions per word.
// Round 1 drew these from Philox4x32-10 (one 4-word block per four positions); the backward's source walks meet forward
// positions in scattered order and had to run the full ten rounds per edge for one word — 2.6-3.2 ms of a 5-9 ms
// walk on the Reddit-sized graph.  Dropout masks need decorrelation, not cryptographic strength.  The (seed, offset)
// pair of a launch goes through splitmix64 ONCE (loop-invariant: the compiler keeps it in scalar registers) to give a
// 64-bit launch key; the word is a two-round keyed mix of the 64-bit index: multiply-xorshift with the key's low half,
// the index's high half and the key's high half folded in before the second multiply, xxHash32's avalanche as the
// finaliser.  Consecutive steps (offset + 1) get unrelated keys in BOTH rounds, so their masks are not related by a
// constant XOR of the pre-avalanche value (round 2's generator: advisor finding), and no index bit is dropped.
// The layer epilogue's dropout (epilogue.hip, reduce.hip) stays on Philox.  Host restatement + the statistical test
// (keep rate, step-to-step and neighbour correlation, bit balance): tests/parity_cases.py gat_drop_word /
// check_drop_word_statistics.
__device__ __forceinline__ uint64_t drop_key(uint64_t offset, uint64_t seed) {
  uint64_t z = seed + offset * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t drop_word(int64_t p, int64_t H, int64_t h, uint64_t offset, uint64_t seed) {
  const uint64_t idx = (uint64_t)(p * H + h);
  const uint64_t key = drop_key(offset, seed);
  uint32_t x = ((uint32_t)idx ^ (uint32_t)key) * 0x9E3779B1u;
  x ^= x >> 15;
  x = (x ^ (uint32_t)(idx >> 32) ^ (uint32_t)(key >> 32)) * 0x85EBCA77u;
  x ^= x >> 13; x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x;
}
// the words of positions 4b .. 4b+3 (the unrolled walks consume them four at a time)
__device__ __forceinline__ U4 drop_words4(int64_t b, int64_t H, int64_t h, uint64_t offset, uint64_t seed) {
  return U4{drop_word(4 * b, H, h, offset, seed), drop_word(4 * b + 1, H, h, offset, seed),
            drop_word(4 * b + 2, H, h, offset, seed), drop_word(4 * b + 3, H, h, offset, seed)};
}

template <int VEC> struct F32V {
  static __device__ __forceinline__ void load(const float *__restrict__ p, float (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = p[i];
  }
  static __device__ __forceinline__ void store(float *__restrict__ p, const float (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = v[i];
  }
};
template <> struct F32V<4> {
  static __device__ __forceinline__ void load(const float *__restrict__ p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float *__restrict__ p, const float (&v)[4]) {
    float4 t;
    t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    *reinterpret_cast<float4 *>(p) = t;
  }
};

static inline int pow2_log2(int64_t v) {
  int l = 0;
  while (l < 6 && ((int64_t)1 << l) < v) ++l;
  return l;
}

static inline int64_t gat_grid_for(int64_t n) {
  int64_t b = ceil_div(n, kBlock);
  if (b > 16384) b = 16384;
  return b < 1 ? 1 : b;
}

static inline int set_dropout(GatDims &d, float p_drop, const int64_t *rng) {
  GGL_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "p_drop must be in [0, 1)");
  GGL_REQUIRE(p_drop == 0.0f || rng, GGL_EINVAL, "attention dropout needs an rng_state");
  d.drop_thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  d.drop_scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  return GGL_OK;
}

// hub-chunk combine kernels (defined in gat.hip)
__global__ void gat_long_final_kernel(const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr,
                                      const float *__restrict__ pacc, const float *__restrict__ pm,
                                      const float *__restrict__ pd, float *__restrict__ y, float *__restrict__ rowmax,
                                      float *__restrict__ rowden, const GatDims d);
__global__ void gat_bwd_dst_final_kernel(const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr,
                                         const float *__restrict__ pger, float *__restrict__ ger, int64_t n_long,
                                         int64_t H);
__global__ void gat_bwd_src_final_kernel(const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr,
                                         const float *__restrict__ pacc, const float *__restrict__ pgel,
                                         float *__restrict__ gx, float *__restrict__ gel, const GatDims d);

}  // namespace ggl
