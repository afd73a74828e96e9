// gammagl_amd/csrc/sample.hip — uniform neighbour sampling on the device.
//
// Supersedes gammagl/ops/sparse `sample_adj` (cpu/sample.cpp:10-135; the reference's GPU sampler is
// ~20 kernels of cuda/neighbor_sample.cu).  Same contract: CSR (rowptr, col) of the in-neighbour lists,
// a batch of seed rows, a fan-out; per seed either every neighbour (fanout < 0), `fanout` draws with
// replacement, or min(deg, fanout) distinct neighbours by Robert Floyd's algorithm — the reference's
// own choice (sample.cpp:75-83): for j = deg - f .. deg - 1: t = randint(j + 1); take t unless already
// taken, else take j.  Randomness: Philox4x32-10 keyed on a device-resident {seed, offset} state and the
// (seed row, draw) pair, so a call is reproducible given the state and independent of scheduling.
// What leaves here is already grouped by seed row, i.e. a CSR block: the segment kernels consume it
// through Engine.plan_from_rowptr with no sort and no host sync (the per-batch plan build measured
// 0.4 ms of a 3.75 ms mini-batch step).  The relabelling of node ids ("first seen" order, seeds first)
// is done by the host layer with device sorts (gammagl_amd/sampler.py).
#include "common.hpp"

namespace ggl {

// Counter = (row a, draw b, call offset): every word of the counter block is its own coordinate, so draws of
// different calls never share a counter (an offset XORed into the draw index made call n, draw j collide with
// call n ^ j, draw 0); the key carries a stream tag, so the sampler's words are also disjoint from the dropout
// masks drawn from the same {seed, offset} state.  a, b < 2^32 (checked at launch).
__device__ __forceinline__ uint32_t philox_u32(uint64_t a, uint64_t b, uint64_t seed, uint64_t offset) {
  uint32_t c0 = (uint32_t)a, c1 = (uint32_t)b, c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
  uint32_t k0 = (uint32_t)seed ^ 0x53414D50u /* 'SAMP' */, k1 = (uint32_t)(seed >> 32);
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
  return c0;
}

// uniform integer in [0, n) from a 32-bit word (multiply-shift; bias < n / 2^32)
__device__ __forceinline__ int64_t bounded(uint32_t r, int64_t n) {
  return (int64_t)(((uint64_t)r * (uint64_t)n) >> 32);
}

__global__ __launch_bounds__(kBlock) void sample_count_kernel(const int64_t *__restrict__ rowptr,
                                                              const int64_t *__restrict__ seeds, int64_t B,
                                                              int64_t fanout, int replace,
                                                              int64_t *__restrict__ out_deg) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < B; i += stride) {
    const int64_t n = seeds[i];
    const int64_t deg = rowptr[n + 1] - rowptr[n];
    int64_t k;
    if (fanout < 0) k = deg;
    else if (replace) k = deg > 0 ? fanout : 0;
    else k = deg < fanout ? deg : fanout;
    out_deg[i] = k;
  }
}

// one thread per seed row that needs a random draw; writes its positions (e_pos = index into col).  Rows
// that keep their whole neighbourhood are left to sample_emit_kernel, which is parallel over OUTPUT positions
// (a thread copying a 100 000-neighbour hub serially made a full-neighbourhood hop 8.7 ms).
__global__ __launch_bounds__(kBlock) void sample_pick_kernel(const int64_t *__restrict__ rowptr,
                                                             const int64_t *__restrict__ col,
                                                             const int64_t *__restrict__ seeds, int64_t B,
                                                             int64_t fanout, int replace,
                                                             const int64_t *__restrict__ out_rowptr,
                                                             const int64_t *__restrict__ rng,
                                                             int64_t *__restrict__ e_pos,
                                                             int64_t *__restrict__ nbr) {
  const uint64_t seed = (uint64_t)rng[0], offset = (uint64_t)rng[1];
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < B; i += stride) {
    const int64_t n = seeds[i];
    const int64_t beg = rowptr[n], deg = rowptr[n + 1] - beg;
    const int64_t o = out_rowptr[i], k = out_rowptr[i + 1] - o;
    if (fanout < 0 || (!replace && deg <= fanout)) {
      continue;  // the whole neighbourhood, in CSR order: sample_emit_kernel
    } else if (replace) {
      for (int64_t j = 0; j < k; ++j) e_pos[o + j] = beg + bounded(philox_u32((uint64_t)i, (uint64_t)j, seed, offset), deg);
    } else {  // Floyd: k = fanout distinct positions out of deg
      for (int64_t j = deg - k, s = 0; j < deg; ++j, ++s) {
        int64_t t = bounded(philox_u32((uint64_t)i, (uint64_t)s, seed, offset), j + 1);
        bool taken = false;
        for (int64_t q = 0; q < s; ++q) taken |= (e_pos[o + q] == beg + t);
        e_pos[o + s] = beg + (taken ? j : t);
      }
    }
  }
}

// one thread per output position q: the seed row that owns q (binary search in out_rowptr), the CSR position
// for rows kept whole, and the neighbour id for every row
__global__ __launch_bounds__(kBlock) void sample_emit_kernel(const int64_t *__restrict__ rowptr,
                                                             const int64_t *__restrict__ col,
                                                             const int64_t *__restrict__ seeds, int64_t B,
                                                             int64_t fanout, int replace,
                                                             const int64_t *__restrict__ out_rowptr,
                                                             int64_t *__restrict__ e_pos,
                                                             int64_t *__restrict__ nbr) {
  const int64_t total = out_rowptr[B];
  const int64_t stride = grid_threads();
  for (int64_t q = thread_id(); q < total; q += stride) {
    int64_t lo = 0, hi = B - 1;  // last i with out_rowptr[i] <= q
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (out_rowptr[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int64_t n = seeds[lo];
    const int64_t beg = rowptr[n], deg = rowptr[n + 1] - beg;
    int64_t pos;
    if (fanout < 0 || (!replace && deg <= fanout)) {
      pos = beg + (q - out_rowptr[lo]);
      e_pos[q] = pos;
    } else {
      pos = e_pos[q];
    }
    nbr[q] = col[pos];
  }
}

__global__ void sample_rng_advance_kernel(int64_t *rng) {
  if (block_id() == 0 && threadIdx.x == 0) rng[1] += 1;
}

static inline int64_t grid_for(int64_t n) {
  int64_t g = ceil_div(n, kBlock);
  if (g > 4096) g = 4096;
  return g < 1 ? 1 : g;
}

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_sample_count(const int64_t *rowptr, const int64_t *seeds, int64_t B, int64_t fanout,
                                int replace, int64_t *out_deg, void *stream) {
  GGL_REQUIRE(B >= 0, GGL_EINVAL, "negative batch");
  if (B == 0) return GGL_OK;
  GGL_REQUIRE(rowptr && seeds && out_deg, GGL_EINVAL, "NULL pointer");
  GGL_LAUNCH((sample_count_kernel), grid_for(B), kBlock, as_stream(stream), rowptr, seeds, B, fanout,
             replace, out_deg);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_sample_pick(const int64_t *rowptr, const int64_t *col, const int64_t *seeds, int64_t B,
                               int64_t fanout, int replace, const int64_t *out_rowptr, int64_t *rng_state,
                               int64_t *e_pos, int64_t *nbr, void *stream) {
  GGL_REQUIRE(B >= 0, GGL_EINVAL, "negative batch");
  if (B == 0) return GGL_OK;
  // col may be NULL for a graph without edges: it is only read at positions the rows own
  GGL_REQUIRE(rowptr && seeds && out_rowptr && rng_state && e_pos && nbr, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(B < ((int64_t)1 << 32) && fanout < ((int64_t)1 << 32), GGL_EINVAL, "batch / fan-out >= 2^32");
  hipStream_t s = as_stream(stream);
  GGL_LAUNCH((sample_pick_kernel), grid_for(B), kBlock, s, rowptr, col, seeds, B, fanout, replace,
             out_rowptr, (const int64_t *)rng_state, e_pos, nbr);
  GGL_LAUNCH_CHECK();
#ifdef GGL_EMULATE
  const int64_t emit_grid = 4;     // grid-stride loop: any grid is correct; the host emulation walks it serially
#else
  const int64_t emit_grid = 4096;  // the output size lives on the device (out_rowptr[B]): fixed grid, strided
#endif
  GGL_LAUNCH((sample_emit_kernel), emit_grid, kBlock, s, rowptr, col, seeds, B, fanout, replace, out_rowptr,
             e_pos, nbr);
  GGL_LAUNCH_CHECK();
  GGL_LAUNCH((sample_rng_advance_kernel), 1, 64, s, rng_state);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}
