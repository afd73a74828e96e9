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

#ifndef GGL_EMULATE
#include <rocprim/rocprim.hpp>
#else
#include <algorithm>
#include <numeric>
#include <vector>
#endif

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

// A seed outside [0, N) is treated as a node without neighbours (no out-of-bounds read of rowptr); its id
// stays in n_id, so the caller's feature gather x[n_id] reports it.
__global__ __launch_bounds__(kBlock) void sample_count_kernel(const int64_t *__restrict__ rowptr,
                                                              const int64_t *__restrict__ seeds, int64_t B,
                                                              int64_t N, int64_t fanout, int replace,
                                                              int64_t *__restrict__ out_deg) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < B; i += stride) {
    const int64_t n = seeds[i];
    const int64_t deg = (n >= 0 && n < N) ? rowptr[n + 1] - rowptr[n] : 0;
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
    const int64_t o = out_rowptr[i], k = out_rowptr[i + 1] - o;
    if (k == 0) continue;  // (also every out-of-range seed: the count kernel gave it no neighbours)
    const int64_t n = seeds[i];
    const int64_t beg = rowptr[n], deg = rowptr[n + 1] - beg;
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

// =====================================================================================================
// One hop with DEVICE-side sizes (ggl_sample_hop): every buffer has a fixed capacity, the number of
// seeds / sampled edges / nodes met lives in device memory and nothing is read back, so sampling, the
// feature gather, both SAGEConv layers, the loss, backward and Adam of a mini-batch step capture into ONE
// hipGraph (the dynamic-shape path above costs two host reads per hop and ~210 launches per mini-batch with
// the GPU idle half of the time).  Relabelling as sample.cpp:24-55,104-130: seeds keep 0..B-1 verbatim
// (duplicates included), new nodes follow in first-seen order, each row's columns ascend by local id.
//   first occurrence of a node = atomicMin of its position into a per-node scratch (order-independent),
//   local ids = exclusive scan over the first-occurrence flags.
// =====================================================================================================
constexpr long long kBigPos = (long long)1 << 62;

__global__ __launch_bounds__(kBlock) void hop_count_kernel(const int64_t *__restrict__ rowptr,
                                                           const int64_t *__restrict__ seeds,
                                                           const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                           int64_t N, int64_t fanout, int64_t *__restrict__ cnt) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i <= B_cap; i += stride) {
    int64_t k = 0;
    if (i < nb) {
      const int64_t s = seeds[i];
      const int64_t deg = (s >= 0 && s < N) ? rowptr[s + 1] - rowptr[s] : 0;  // out-of-range seed: no neighbours
      k = deg < fanout ? deg : fanout;
    }
    cnt[i] = k;  // cnt[B_cap] = 0: the exclusive scan leaves the total there
  }
}

// thread per seed row: Floyd's algorithm (sample.cpp:75-83) or the whole neighbourhood when deg <= fanout
__global__ __launch_bounds__(kBlock) void hop_pick_kernel(const int64_t *__restrict__ rowptr,
                                                          const int64_t *__restrict__ col,
                                                          const int64_t *__restrict__ seeds,
                                                          const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                          int64_t fanout, const int64_t *__restrict__ out_rowptr,
                                                          const int64_t *__restrict__ rng, int64_t *__restrict__ e_pos,
                                                          int64_t *__restrict__ nbr) {
  const uint64_t seed = (uint64_t)rng[0], offset = (uint64_t)rng[1];
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < nb; i += stride) {
    const int64_t o = out_rowptr[i], k = out_rowptr[i + 1] - o;
    if (k == 0) continue;
    const int64_t n = seeds[i];
    const int64_t beg = rowptr[n], deg = rowptr[n + 1] - beg;
    if (deg <= fanout) {
      for (int64_t j = 0; j < k; ++j) e_pos[o + j] = beg + j;
    } else {
      for (int64_t j = deg - k, s = 0; j < deg; ++j, ++s) {
        const int64_t t = bounded(philox_u32((uint64_t)i, (uint64_t)s, seed, offset), j + 1);
        bool taken = false;
        for (int64_t q = 0; q < s; ++q) taken |= (e_pos[o + q] == beg + t);
        e_pos[o + s] = beg + (taken ? j : t);
      }
    }
    for (int64_t j = 0; j < k; ++j) nbr[o + j] = col[e_pos[o + j]];
  }
}

// keys: seed i -> -(i + 1), sampled neighbour q -> B_cap + q.  The minimum per node is the LAST seed position
// when the node is a seed (a seed listed twice maps to its last position: operator[] overwrite, sample.cpp:27)
// and otherwise the first sampled occurrence.
__global__ __launch_bounds__(kBlock) void hop_mark_kernel(const int64_t *__restrict__ seeds,
                                                          const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                          const int64_t *__restrict__ nbr,
                                                          const int64_t *__restrict__ out_rowptr, int64_t N,
                                                          long long *__restrict__ first_pos) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t ne = out_rowptr[B_cap];
  const int64_t stride = grid_threads();
  for (int64_t t = thread_id(); t < nb + ne; t += stride) {
    if (t < nb) {
      const int64_t sd = seeds[t];
      if (sd >= 0 && sd < N) atomicMin(&first_pos[sd], -(long long)(t + 1));
    } else {
      atomicMin(&first_pos[nbr[t - nb]], (long long)(B_cap + (t - nb)));
    }
  }
}

// flag[pos] = this position introduces a node: every seed (verbatim, sample.cpp:24-29) and the first
// occurrence of every node that is not a seed
__global__ __launch_bounds__(kBlock) void hop_flag_kernel(const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                          int64_t E_cap, const int64_t *__restrict__ nbr,
                                                          const int64_t *__restrict__ out_rowptr,
                                                          const long long *__restrict__ first_pos,
                                                          int64_t *__restrict__ flag) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t ne = out_rowptr[B_cap];
  const int64_t stride = grid_threads();
  for (int64_t t = thread_id(); t <= B_cap + E_cap; t += stride) {
    int64_t f = 0;
    if (t < B_cap) f = t < nb ? 1 : 0;
    else if (t < B_cap + E_cap) {
      const int64_t q = t - B_cap;
      f = (q < ne && first_pos[nbr[q]] == (long long)t) ? 1 : 0;
    }
    flag[t] = f;  // flag[B_cap + E_cap] = 0: the exclusive scan leaves the node count there
  }
}

__global__ __launch_bounds__(kBlock) void hop_emit_kernel(const int64_t *__restrict__ seeds,
                                                          const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                          int64_t E_cap, const int64_t *__restrict__ nbr,
                                                          const int64_t *__restrict__ out_rowptr,
                                                          const long long *__restrict__ first_pos,
                                                          const int64_t *__restrict__ flag,
                                                          const int64_t *__restrict__ new_id,
                                                          int64_t S_cap, int64_t *__restrict__ out_nid,
                                                          int64_t *__restrict__ local,
                                                          int64_t *__restrict__ counts,
                                                          int64_t *__restrict__ overflow_total) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t ne = out_rowptr[B_cap];
  const int64_t n_all = new_id[B_cap + E_cap];
  const int64_t n_nodes = n_all < S_cap ? n_all : S_cap;   // S_cap >= B_cap: the seeds always fit
  const int64_t stride = grid_threads();
  if (thread_id() == 0) {
    counts[0] = n_nodes;
    counts[1] = ne;
    const int64_t over = (n_all > S_cap || counts[2] != 0) ? 1 : 0;   // (counts[2]: rows cut at E_cap, set by the first scan)
    if (n_all > S_cap) counts[2] = 1;  // more nodes met than the caller's capacity: the block is truncated
    // the caller's running count of hops that hit a capacity (BlockSampler.overflow_count): was a torch add per hop
    if (overflow_total != nullptr) *overflow_total += over;
  }
  for (int64_t t = thread_id(); t < S_cap; t += stride) {
    if (t >= n_nodes) out_nid[t] = 0;  // padding rows gather node 0 (any valid row: nothing reads them)
  }
  for (int64_t t = thread_id(); t < B_cap + E_cap; t += stride) {
    if (t < B_cap) {
      if (t < nb) out_nid[new_id[t]] = seeds[t];
    } else {
      const int64_t q = t - B_cap;
      if (q < ne) {
        const int64_t node = nbr[q];
        if (flag[t] && new_id[t] < S_cap) out_nid[new_id[t]] = node;
        const long long key = first_pos[node];
        const int64_t l = key < 0 ? (int64_t)(-key - 1) : new_id[key];  // a seed keeps its own (last) position
        local[q] = l < S_cap ? l : S_cap - 1;  // (only on overflow, which the caller must check)
      } else {
        local[q] = 0;
      }
    }
  }
}

// after the scan: rows that would run past the edge capacity are cut (and the overflow flag raised)
__global__ __launch_bounds__(kBlock) void hop_clamp_kernel(int64_t *__restrict__ out_rowptr, int64_t B_cap,
                                                           int64_t E_cap, int64_t *__restrict__ counts) {
  const int64_t stride = grid_threads();
  const bool over = out_rowptr[B_cap] > E_cap;  // (read before any thread of this launch clamps it: see below)
  for (int64_t i = thread_id(); i < B_cap; i += stride)
    if (out_rowptr[i] > E_cap) out_rowptr[i] = E_cap;
  if (thread_id() == 0) counts[2] = over ? 1 : 0;
}
__global__ void hop_clamp_last_kernel(int64_t *__restrict__ out_rowptr, int64_t B_cap, int64_t E_cap) {
  if (block_id() == 0 && threadIdx.x == 0 && out_rowptr[B_cap] > E_cap) out_rowptr[B_cap] = E_cap;
}
// (out_nid is written by two loops of the same launch: a padding slot j >= n_nodes is never a new_id target,
//  so the two never touch the same element)

__global__ __launch_bounds__(kBlock) void hop_reset_kernel(const int64_t *__restrict__ seeds,
                                                           const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                           const int64_t *__restrict__ nbr,
                                                           const int64_t *__restrict__ out_rowptr, int64_t N,
                                                           long long *__restrict__ first_pos, int64_t *__restrict__ rng) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t ne = out_rowptr[B_cap];
  const int64_t stride = grid_threads();
  for (int64_t t = thread_id(); t < nb + ne; t += stride) {
    const int64_t node = t < nb ? seeds[t] : nbr[t - nb];
    if (node >= 0 && node < N) first_pos[node] = kBigPos;
  }
  // the hop's draws are done (hop_pick ran earlier on this stream): the next hop gets a fresh offset — was a launch
  // of its own (sample_rng_advance_kernel)
  if (rng != nullptr && thread_id() == 0) rng[1] += 1;
}

// thread per row: columns ascending by local id (sample.cpp:112-118), e_pos carried along; rows hold <= fanout
// entries, so an in-place insertion sort is a handful of steps
__global__ __launch_bounds__(kBlock) void hop_rowsort_kernel(const int64_t *__restrict__ out_rowptr, int64_t B_cap,
                                                             int64_t E_cap, int64_t *__restrict__ local,
                                                             int64_t *__restrict__ e_pos,
                                                             int32_t *__restrict__ out_col,
                                                             int64_t *__restrict__ out_eid) {
  const int64_t ne = out_rowptr[B_cap];
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < B_cap; i += stride) {
    const int64_t b = out_rowptr[i], e = out_rowptr[i + 1];
    for (int64_t a = b + 1; a < e; ++a) {
      const int64_t lv = local[a], ev = e_pos[a];
      int64_t c = a - 1;
      while (c >= b && local[c] > lv) {
        local[c + 1] = local[c];
        e_pos[c + 1] = e_pos[c];
        --c;
      }
      local[c + 1] = lv;
      e_pos[c + 1] = ev;
    }
    for (int64_t a = b; a < e; ++a) {
      out_col[a] = (int32_t)local[a];
      if (out_eid) out_eid[a] = e_pos[a];
    }
  }
  for (int64_t q = ne + thread_id(); q < E_cap; q += stride) {  // padding past the sampled edges
    out_col[q] = 0;
    if (out_eid) out_eid[q] = 0;
  }
}


// ---- register-resident variants for fan-outs <= 32 (the reference's [25, 10]) ---------------------------------
// hop_pick_kernel walks Floyd's algorithm with its picks in global memory (an O(k^2) chain of dependent global
// reads per thread: 46 us per hop on the products-sized graph) and hop_rowsort_kernel insertion-sorts rows in
// global memory (61 us); with the row in registers both are a few microseconds: the random draws of Floyd's steps do
// not depend on earlier picks (only the membership test does), and a 32-element bitonic network sorts a row.
constexpr int kRegF = 32;

__global__ __launch_bounds__(kBlock) void hop_pick_reg_kernel(const int64_t *__restrict__ rowptr,
                                                              const int64_t *__restrict__ col,
                                                              const int64_t *__restrict__ seeds,
                                                              const int64_t *__restrict__ n_seeds, int64_t B_cap,
                                                              int64_t fanout, const int64_t *__restrict__ out_rowptr,
                                                              const int64_t *__restrict__ rng, int64_t *__restrict__ e_pos,
                                                              int64_t *__restrict__ nbr) {
  const uint64_t seed = (uint64_t)rng[0], offset = (uint64_t)rng[1];
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < nb; i += stride) {
    const int64_t o = out_rowptr[i];
    const int k = (int)(out_rowptr[i + 1] - o);
    if (k == 0) continue;
    const int64_t n = seeds[i];
    const int64_t beg = rowptr[n], deg = rowptr[n + 1] - beg;
    int64_t pk[kRegF];
    if (deg <= fanout) {
#pragma unroll
      for (int s = 0; s < kRegF; ++s) pk[s] = s;
    } else {
      const int64_t j0 = deg - k;
#pragma unroll
      for (int s = 0; s < kRegF; ++s)  // the draws are independent of the earlier picks
        pk[s] = s < k ? bounded(philox_u32((uint64_t)i, (uint64_t)s, seed, offset), j0 + s + 1) : -1 - s;
#pragma unroll
      for (int s = 1; s < kRegF; ++s) {  // resolve in order: a draw already taken is replaced by j (Floyd)
        bool taken = false;
#pragma unroll
        for (int q = 0; q < s; ++q) taken |= (pk[q] == pk[s]);
        if (s < k && taken) pk[s] = j0 + s;
      }
    }
    int64_t nv[kRegF];
#pragma unroll
    for (int s = 0; s < kRegF; ++s) nv[s] = s < k ? col[beg + pk[s]] : 0;  // k independent gathers
#pragma unroll
    for (int s = 0; s < kRegF; ++s) {
      if (s < k) {
        e_pos[o + s] = beg + pk[s];
        nbr[o + s] = nv[s];
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void hop_rowsort_reg_kernel(const int64_t *__restrict__ out_rowptr, int64_t B_cap,
                                                                 int64_t E_cap, const int64_t *__restrict__ local,
                                                                 const int64_t *__restrict__ e_pos,
                                                                 int32_t *__restrict__ out_col,
                                                                 int64_t *__restrict__ out_eid) {
  const int64_t ne = out_rowptr[B_cap];
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < B_cap; i += stride) {
    const int64_t b = out_rowptr[i];
    const int k = (int)(out_rowptr[i + 1] - b);
    if (k == 0) continue;
    int32_t key[kRegF];
    int64_t val[kRegF];
#pragma unroll
    for (int s = 0; s < kRegF; ++s) {
      key[s] = s < k ? (int32_t)local[b + s] : INT32_MAX;  // padding sorts to the end
      val[s] = s < k ? e_pos[b + s] : 0;
    }
    // bitonic network on 32 (key, val) pairs; keys within a row are distinct (distinct neighbours), so the
    // order is total and equals the insertion sort's
#pragma unroll
    for (int size = 2; size <= kRegF; size <<= 1) {
#pragma unroll
      for (int st = size >> 1; st > 0; st >>= 1) {
#pragma unroll
        for (int a = 0; a < kRegF; ++a) {
          const int c = a ^ st;
          if (c > a) {
            const bool up = (a & size) == 0;
            const bool sw = up ? key[a] > key[c] : key[a] < key[c];
            const int32_t ka = key[a], kc = key[c];
            const int64_t va = val[a], vc = val[c];
            key[a] = sw ? kc : ka; key[c] = sw ? ka : kc;
            val[a] = sw ? vc : va; val[c] = sw ? va : vc;
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kRegF; ++s) {
      if (s < k) {
        out_col[b + s] = key[s];
        if (out_eid) out_eid[b + s] = val[s];
      }
    }
  }
  for (int64_t q = ne + thread_id(); q < E_cap; q += stride) {  // padding past the sampled edges
    out_col[q] = 0;
    if (out_eid) out_eid[q] = 0;
  }
}

// ---- transposed structure of a block, asynchronously (for the backward of its aggregate) --------------
// key[q] = source column of edge q (N_src_cap for the padding past n_edges: sorts to the end),
// val[q] = destination row of q (binary search in rowptr)
__global__ __launch_bounds__(kBlock) void block_keys_kernel(const int64_t *__restrict__ rowptr, int64_t N_dst,
                                                            const int32_t *__restrict__ col, int64_t E_cap,
                                                            int64_t N_src_cap, uint32_t *__restrict__ keys,
                                                            int32_t *__restrict__ vals) {
  const int64_t ne = rowptr[N_dst];
  const int64_t stride = grid_threads();
  for (int64_t q = thread_id(); q < E_cap; q += stride) {
    if (q >= ne) {
      keys[q] = (uint32_t)N_src_cap;
      vals[q] = 0;
      continue;
    }
    int64_t lo = 0, hi = N_dst;  // first r with rowptr[r + 1] > q
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid + 1] <= q) lo = mid + 1; else hi = mid;
    }
    keys[q] = (uint32_t)col[q];
    vals[q] = (int32_t)lo;
  }
}

__global__ __launch_bounds__(kBlock) void block_rowptrT_kernel(const uint32_t *__restrict__ keys, int64_t E_cap,
                                                               int64_t N_src_cap, int64_t *__restrict__ rowptrT) {
  const int64_t stride = grid_threads();
  for (int64_t s = thread_id(); s <= N_src_cap; s += stride) {
    int64_t lo = 0, hi = E_cap;  // first position with key >= s
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < s) lo = mid + 1; else hi = mid;
    }
    rowptrT[s] = lo;
  }
}

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
static inline int bits_for(int64_t n) {
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) <= n) ++b;
  return b;
}
#ifndef GGL_EMULATE
static size_t scan_temp_bytes(int64_t n) {
  size_t tmp = 0;
  (void)rocprim::exclusive_scan(nullptr, tmp, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n,
                                rocprim::plus<int64_t>(), (hipStream_t)0);
  return tmp;
}
static size_t sortpairs_temp_bytes(int64_t n, int bits) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)nullptr,
                                  (int32_t *)nullptr, (size_t)n, 0u, (unsigned)bits, (hipStream_t)0);
  return tmp;
}
#endif
static int scan_i64(void *tmp, size_t tmp_bytes, const int64_t *in, int64_t *out, int64_t n, hipStream_t s) {
#ifndef GGL_EMULATE
  GGL_HIP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), s));
#else
  (void)tmp; (void)tmp_bytes; (void)s;
  int64_t acc = 0;
  for (int64_t i = 0; i < n; ++i) { const int64_t v = in[i]; out[i] = acc; acc += v; }
#endif
  return GGL_OK;
}

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_sample_count(const int64_t *rowptr, const int64_t *seeds, int64_t B, int64_t num_nodes,
                                int64_t fanout, int replace, int64_t *out_deg, void *stream) {
  GGL_REQUIRE(B >= 0 && num_nodes >= 0, GGL_EINVAL, "negative size");
  if (B == 0) return GGL_OK;
  GGL_REQUIRE(rowptr && seeds && out_deg, GGL_EINVAL, "NULL pointer");
  GGL_LAUNCH((sample_count_kernel), grid_for(B), kBlock, as_stream(stream), rowptr, seeds, B, num_nodes, fanout,
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

#ifndef GGL_EMULATE
// ---- count / flag + exclusive scan in ONE launch each (round 5) ------------------------------------------------------
// The hop's two scans ran as [produce values] + rocprim's [init look-back state] + [scan] (+ two clamp kernels after the first):
// 8 of the hop's 14 launches, ~5 us each in a replayed step whose kernels take less than that.  Here the values are computed
// inside a single-pass chained scan: a block scans its 2048 values, publishes its total, and adds up its predecessors' by
// looking back (decoupled look-back: a block waits only for blocks with lower ids, which were dispatched before it).
// The look-back state needs no initialisation launch: every slot carries a 64-bit TAG made of the hop's rng offset (which the
// hop's last kernel increments: unique per hop, also across replays of a recorded step), the scan's id and the block — a
// slot is read only when its tag matches, whatever the buffer held before.
struct ScanSlot {
  unsigned long long tag;   // written LAST (release): the slot belongs to this hop, this scan, this block
  unsigned long long fv;    // (flag << 32) | value; flag 1 = the block's own total, 2 = the total of blocks 0 .. b
};
constexpr int kScanIpt = 8, kScanTile = kBlock * kScanIpt;   // values per thread / per block
__device__ __forceinline__ unsigned long long scan_tag(unsigned long long base, int64_t b) {
  unsigned long long z = base + 0x9E3779B97F4A7C15ull * (unsigned long long)(b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (z ^ (z >> 31)) | 1ull;
}
// exclusive prefix of this thread's partial sum `tsum` over the whole grid (thread order = block-major): returns the sum of
// everything before this thread.  Every thread of every block must call it once.
__device__ __forceinline__ uint32_t chained_exclusive(uint32_t tsum, ScanSlot *__restrict__ state, unsigned long long base) {
  __shared__ uint32_t wsum[kWavesPerBlock];
  __shared__ uint32_t bprefix;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  uint32_t incl = tsum;
#pragma unroll
  for (int dlt = 1; dlt < kWave; dlt <<= 1) {
    const uint32_t y = __shfl_up(incl, dlt, kWave);
    if (lane >= dlt) incl += y;
  }
  if (lane == kWave - 1) wsum[wave] = incl;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kWavesPerBlock; ++q) {
    if (q < wave) woff += wsum[q];
    total += wsum[q];
  }
  if (threadIdx.x == 0) {
    const int64_t b = block_id();
    ScanSlot *mine = state + b;
    uint32_t run = 0;
    if (b == 0) {
      __hip_atomic_store(&mine->fv, (2ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine->tag, scan_tag(base, b), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store(&mine->fv, (1ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine->tag, scan_tag(base, b), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      for (int64_t j = b - 1; j >= 0; --j) {
        const unsigned long long want = scan_tag(base, j);
        while (__hip_atomic_load(&state[j].tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
        const unsigned long long fv = __hip_atomic_load(&state[j].fv, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        run += (uint32_t)fv;
        if ((fv >> 32) == 2ull) break;
      }
      __hip_atomic_store(&mine->fv, (2ull << 32) | (unsigned long long)(run + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    bprefix = run;
  }
  __syncthreads();
  return bprefix + woff + (incl - tsum);
}

// hop_count_kernel + exclusive scan + hop_clamp_kernel + hop_clamp_last_kernel: out_rowptr[i] = min(E_cap, sum of the first i
// rows' sample counts), i in [0, B_cap]; counts[2] = 1 when the unclamped total exceeds E_cap
__global__ __launch_bounds__(kBlock) void hop_count_scan_kernel(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ seeds,
                                                                const int64_t *__restrict__ n_seeds, int64_t B_cap, int64_t N,
                                                                int64_t fanout, int64_t E_cap, const int64_t *__restrict__ rng,
                                                                ScanSlot *__restrict__ state, int64_t *__restrict__ out_rowptr,
                                                                int64_t *__restrict__ counts) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t i0 = block_id() * kScanTile + (int64_t)threadIdx.x * kScanIpt;
  uint32_t v[kScanIpt], tsum = 0;
#pragma unroll
  for (int q = 0; q < kScanIpt; ++q) {
    const int64_t i = i0 + q;
    int64_t k = 0;
    if (i < nb) {
      const int64_t sd = seeds[i];
      const int64_t deg = (sd >= 0 && sd < N) ? rowptr[sd + 1] - rowptr[sd] : 0;
      k = deg < fanout ? deg : fanout;
    }
    v[q] = (uint32_t)k;
    tsum += v[q];
  }
  // (tag base = scan id + the stream's offset AND its seed: after a reseed / state restore the offset repeats, and the caching
  //  allocator hands back the same workspace — a tag without the seed would accept a stale predecessor slot; round-5 advisor)
  uint64_t p = chained_exclusive(tsum, state, ((((unsigned long long)rng[1] << 2) | 1ull) ^ ((unsigned long long)rng[0] * 0xD6E8FEB86659FD93ull)));
#pragma unroll
  for (int q = 0; q < kScanIpt; ++q) {
    const int64_t i = i0 + q;
    if (i <= B_cap) {
      out_rowptr[i] = (int64_t)p < E_cap ? (int64_t)p : E_cap;
      if (i == B_cap) counts[2] = (int64_t)p > E_cap ? 1 : 0;
    }
    p += v[q];
  }
}

// hop_flag_kernel + exclusive scan: flag[t] (this position introduces a node) and new_id[t] (nodes introduced before it),
// t in [0, B_cap + E_cap]
__global__ __launch_bounds__(kBlock) void hop_flag_scan_kernel(const int64_t *__restrict__ n_seeds, int64_t B_cap, int64_t E_cap,
                                                               const int64_t *__restrict__ nbr, const int64_t *__restrict__ out_rowptr,
                                                               const long long *__restrict__ first_pos, const int64_t *__restrict__ rng,
                                                               ScanSlot *__restrict__ state, int64_t *__restrict__ flag,
                                                               int64_t *__restrict__ new_id) {
  const int64_t nb = *n_seeds < B_cap ? *n_seeds : B_cap;
  const int64_t ne = out_rowptr[B_cap];
  const int64_t T = B_cap + E_cap;
  const int64_t i0 = block_id() * kScanTile + (int64_t)threadIdx.x * kScanIpt;
  uint32_t v[kScanIpt], tsum = 0;
#pragma unroll
  for (int q = 0; q < kScanIpt; ++q) {
    const int64_t t = i0 + q;
    uint32_t f = 0;
    if (t < B_cap) f = t < nb ? 1u : 0u;
    else if (t < T) {
      const int64_t e = t - B_cap;
      f = (e < ne && first_pos[nbr[e]] == (long long)t) ? 1u : 0u;
    }
    v[q] = f;
    tsum += f;
  }
  uint32_t p = chained_exclusive(tsum, state, ((((unsigned long long)rng[1] << 2) | 2ull) ^ ((unsigned long long)rng[0] * 0xD6E8FEB86659FD93ull)));
#pragma unroll
  for (int q = 0; q < kScanIpt; ++q) {
    const int64_t t = i0 + q;
    if (t <= T) {
      flag[t] = (int64_t)v[q];
      new_id[t] = (int64_t)p;
    }
    p += v[q];
  }
}
#endif

#ifndef GGL_EMULATE
// ---- small hops: scan + clamp (+ clamp_last), and the flag scan, as ONE single-workgroup launch each (round 6) -----------------------
// A replayed mini-batch step is bound by the number of its dependent ~5-us launches.  The hop's first scan runs over B_cap + 1 <= 2^16
// values — rocprim's [init] + [scan] + [clamp] + [clamp_last] = four launches, 20 us, for 150 KB of work that one workgroup of 1024
// threads finishes in one launch: no inter-workgroup state, no look-back chain (what made round 5's multi-block fused scans as slow as
// the launches they replaced).  The VALUES stay with their parallel kernels (hop_count / hop_flag): computed inside the single
// workgroup they cost 2-50 dependent random reads per thread on ONE CU (measured: the step 0.62 -> 0.76 ms).  The second scan (flags
// over B_cap + E_cap + 1 positions) takes the same form only while it is small: one CU streams ~130 GB/s, rocprim's two launches take 10 us.
constexpr int kSmallThreads = 1024, kSmallWaves = kSmallThreads / 64;
constexpr int64_t kSmallScanMax = 65536;       // values the first scan's single launch covers
constexpr int64_t kSmallFlagMax = 32768;       // ... and the flag scan's
// out[i] = min(cap, sum of in[0 .. i)) for i in [0, n); over_flag (or NULL) = the unclamped total exceeds cap (cap < 0: no clamp).
// = exclusive scan + hop_clamp_kernel + hop_clamp_last_kernel.  Wavefront w owns the contiguous slice [w C, (w + 1) C) and walks it in
// COALESCED 64-element steps (a lane per element, a shuffle scan per step, the running prefix carried in a register): pass 1 the slices'
// totals, one barrier, pass 2 the prefixes.  (A thread-contiguous split reads 64 different lines per load instruction: measured slower
// than the four launches it replaces.)
__global__ __launch_bounds__(kSmallThreads) void scan_clamp_small_kernel(const int64_t *__restrict__ in, int64_t n, int64_t cap,
                                                                        int64_t *__restrict__ out, int64_t *__restrict__ over_flag) {
  __shared__ uint64_t wsum[kSmallWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t C = ((n + kSmallWaves - 1) / kSmallWaves + 63) / 64 * 64;     // slice length: a multiple of the step
  const int64_t lo = (int64_t)wave * C, hi = lo + C < n ? lo + C : n;
  uint64_t tot = 0;
  for (int64_t i = lo + lane; i < hi; i += 64) tot += (uint64_t)in[i];
#pragma unroll
  for (int dlt = 32; dlt > 0; dlt >>= 1) tot += __shfl_xor(tot, dlt, 64);
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  uint64_t run = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kSmallWaves; ++q) {
    if (q < wave) run += wsum[q];
    total += wsum[q];
  }
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t i = base + lane;
    const uint64_t v = i < hi ? (uint64_t)in[i] : 0ull;      // (read before out[i] is written: in and out may be the same array)
    uint64_t incl = v;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint64_t y = __shfl_up(incl, dlt, 64);
      if (lane >= dlt) incl += y;
    }
    const uint64_t p = run + incl - v;
    if (i < hi) out[i] = (cap >= 0 && (int64_t)p > cap) ? cap : (int64_t)p;
    run += __shfl(incl, 63, 64);
  }
  if (over_flag && threadIdx.x == 0) *over_flag = (cap >= 0 && (int64_t)total > cap) ? 1 : 0;
}
#endif

// ---- static-shape hop -------------------------------------------------------------------------------
extern "C" size_t ggl_sample_hop_workspace_bytes(int64_t B_cap, int64_t E_cap) {
  if (B_cap < 0 || E_cap < 0) return 0;
  const int64_t T = B_cap + E_cap + 1;
  size_t b = up256((size_t)(B_cap + 1) * 8);          // cnt
  b += 3 * up256((size_t)(E_cap > 0 ? E_cap : 1) * 8);  // e_pos, nbr, local
  b += 2 * up256((size_t)T * 8);                      // flag, new_id
#ifndef GGL_EMULATE
  b += up256(scan_temp_bytes(T > B_cap + 1 ? T : B_cap + 1));
  b += 2 * up256((size_t)(ceil_div(T + 1, (int64_t)kScanTile) + 1) * sizeof(ScanSlot));   // look-back slots of the two fused scans
#endif
  return b + 256;
}

extern "C" int ggl_sample_hop_ex(const int64_t *rowptr, const int64_t *col, const int64_t *seeds,
                                 const int64_t *n_seeds_dev, int64_t B_cap, int64_t num_nodes, int64_t fanout,
                                 int64_t E_cap, int64_t S_cap, int64_t *rng_state, int64_t *first_pos, int64_t *out_rowptr,
                                 int32_t *out_col, int64_t *out_eid, int64_t *out_nid, int64_t *out_counts,
                                 void *workspace, size_t workspace_bytes, int64_t *overflow_total, void *stream) {
  GGL_REQUIRE(B_cap >= 0 && fanout > 0, GGL_EINVAL, "the static-shape hop needs a positive fan-out");
  GGL_REQUIRE(E_cap > 0 && E_cap <= B_cap * fanout && S_cap >= B_cap && S_cap <= B_cap + E_cap, GGL_EINVAL,
              "capacities: 0 < E_cap <= B_cap * fanout, B_cap <= S_cap <= B_cap + E_cap");
  GGL_REQUIRE(B_cap < ((int64_t)1 << 31) && E_cap < ((int64_t)1 << 31), GGL_EINVAL, "block too large");
  if (B_cap == 0) return GGL_OK;
  GGL_REQUIRE(rowptr && seeds && n_seeds_dev && rng_state && first_pos && out_rowptr && out_col && out_nid &&
                  out_counts, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_sample_hop_workspace_bytes(B_cap, E_cap), GGL_EWORKSPACE,
              "sample_hop workspace too small");
  hipStream_t s = as_stream(stream);
  const int64_t T = B_cap + E_cap + 1;
  char *ws = static_cast<char *>(workspace);
  size_t off = 0;
  int64_t *cnt = reinterpret_cast<int64_t *>(ws + off); off += up256((size_t)(B_cap + 1) * 8);
  int64_t *e_pos = reinterpret_cast<int64_t *>(ws + off); off += up256((size_t)E_cap * 8);
  int64_t *nbr = reinterpret_cast<int64_t *>(ws + off); off += up256((size_t)E_cap * 8);
  int64_t *local = reinterpret_cast<int64_t *>(ws + off); off += up256((size_t)E_cap * 8);
  int64_t *flag = reinterpret_cast<int64_t *>(ws + off); off += up256((size_t)T * 8);
  int64_t *new_id = reinterpret_cast<int64_t *>(ws + off); off += up256((size_t)T * 8);
  void *tmp = ws + off;
  const size_t tmp_bytes = workspace_bytes - off;
  long long *fp = reinterpret_cast<long long *>(first_pos);
  int rc = GGL_OK;
#ifndef GGL_EMULATE
  // the look-back slots of the two fused scans sit behind rocprim's scratch (sized in ggl_sample_hop_workspace_bytes)
  const size_t slots_bytes = up256((size_t)(ceil_div(T + 1, (int64_t)kScanTile) + 1) * sizeof(ScanSlot));
  const size_t rp_bytes = up256(scan_temp_bytes(T > B_cap + 1 ? T : B_cap + 1));
  ScanSlot *slots1 = reinterpret_cast<ScanSlot *>(ws + off + rp_bytes);
  ScanSlot *slots2 = reinterpret_cast<ScanSlot *>(ws + off + rp_bytes + slots_bytes);
  const bool fused_scans = options().hop_fused_scans != 0;
#else
  const bool fused_scans = false;
#endif
#ifndef GGL_EMULATE
  const bool small1 = options().hop_small_scans != 0 && B_cap + 1 <= kSmallScanMax;
  const bool small2 = options().hop_small_scans != 0 && T <= kSmallFlagMax;
#else
  const bool small1 = false, small2 = false;
#endif
  if (small1) {
#ifndef GGL_EMULATE
    GGL_LAUNCH((hop_count_kernel), grid_for(B_cap + 1), kBlock, s, rowptr, seeds, n_seeds_dev, B_cap, num_nodes, fanout, cnt);
    GGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_clamp_small_kernel, dim3(1), dim3(kSmallThreads), 0, s, (const int64_t *)cnt, B_cap + 1, E_cap, out_rowptr,
                       out_counts + 2);
    GGL_LAUNCH_CHECK();
#endif
  } else if (fused_scans) {
#ifndef GGL_EMULATE
    GGL_LAUNCH((hop_count_scan_kernel), ceil_div(B_cap + 1, (int64_t)kScanTile), kBlock, s, rowptr, seeds, n_seeds_dev, B_cap,
               num_nodes, fanout, E_cap, (const int64_t *)rng_state, slots1, out_rowptr, out_counts);
    GGL_LAUNCH_CHECK();
#endif
  } else {
    GGL_LAUNCH((hop_count_kernel), grid_for(B_cap + 1), kBlock, s, rowptr, seeds, n_seeds_dev, B_cap, num_nodes, fanout,
               cnt);
    GGL_LAUNCH_CHECK();
    rc = scan_i64(tmp, tmp_bytes, cnt, out_rowptr, B_cap + 1, s);
    if (rc) return rc;
    GGL_LAUNCH((hop_clamp_kernel), grid_for(B_cap), kBlock, s, out_rowptr, B_cap, E_cap, out_counts);
    GGL_LAUNCH_CHECK();
    GGL_LAUNCH((hop_clamp_last_kernel), 1, 64, s, out_rowptr, B_cap, E_cap);
    GGL_LAUNCH_CHECK();
  }
  const bool in_regs = fanout <= kRegF;
  if (in_regs)
    GGL_LAUNCH((hop_pick_reg_kernel), grid_for(B_cap), kBlock, s, rowptr, col, seeds, n_seeds_dev, B_cap, fanout,
               (const int64_t *)out_rowptr, (const int64_t *)rng_state, e_pos, nbr);
  else
    GGL_LAUNCH((hop_pick_kernel), grid_for(B_cap), kBlock, s, rowptr, col, seeds, n_seeds_dev, B_cap, fanout,
               (const int64_t *)out_rowptr, (const int64_t *)rng_state, e_pos, nbr);
  GGL_LAUNCH_CHECK();
  GGL_LAUNCH((hop_mark_kernel), grid_for(B_cap + E_cap), kBlock, s, seeds, n_seeds_dev, B_cap, (const int64_t *)nbr,
             (const int64_t *)out_rowptr, num_nodes, fp);
  GGL_LAUNCH_CHECK();
  if (small2) {
#ifndef GGL_EMULATE
    GGL_LAUNCH((hop_flag_kernel), grid_for(T), kBlock, s, n_seeds_dev, B_cap, E_cap, (const int64_t *)nbr,
               (const int64_t *)out_rowptr, (const long long *)fp, flag);
    GGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_clamp_small_kernel, dim3(1), dim3(kSmallThreads), 0, s, (const int64_t *)flag, T, (int64_t)-1, new_id,
                       (int64_t *)nullptr);
    GGL_LAUNCH_CHECK();
#endif
  } else if (fused_scans) {
#ifndef GGL_EMULATE
    GGL_LAUNCH((hop_flag_scan_kernel), ceil_div(T, (int64_t)kScanTile), kBlock, s, n_seeds_dev, B_cap, E_cap, (const int64_t *)nbr,
               (const int64_t *)out_rowptr, (const long long *)fp, (const int64_t *)rng_state, slots2, flag, new_id);
    GGL_LAUNCH_CHECK();
#endif
  } else {
    GGL_LAUNCH((hop_flag_kernel), grid_for(T), kBlock, s, n_seeds_dev, B_cap, E_cap, (const int64_t *)nbr,
               (const int64_t *)out_rowptr, (const long long *)fp, flag);
    GGL_LAUNCH_CHECK();
    rc = scan_i64(tmp, tmp_bytes, flag, new_id, T, s);
    if (rc) return rc;
  }
  GGL_LAUNCH((hop_emit_kernel), grid_for(B_cap + E_cap), kBlock, s, seeds, n_seeds_dev, B_cap, E_cap,
             (const int64_t *)nbr, (const int64_t *)out_rowptr, (const long long *)fp, (const int64_t *)flag,
             (const int64_t *)new_id, S_cap, out_nid, local, out_counts, overflow_total);
  GGL_LAUNCH_CHECK();
  GGL_LAUNCH((hop_reset_kernel), grid_for(B_cap + E_cap), kBlock, s, seeds, n_seeds_dev, B_cap, (const int64_t *)nbr,
             (const int64_t *)out_rowptr, num_nodes, fp, rng_state);
  GGL_LAUNCH_CHECK();
  if (in_regs)
    GGL_LAUNCH((hop_rowsort_reg_kernel), grid_for(B_cap), kBlock, s, (const int64_t *)out_rowptr, B_cap, E_cap,
               (const int64_t *)local, (const int64_t *)e_pos, out_col, out_eid);
  else
    GGL_LAUNCH((hop_rowsort_kernel), grid_for(B_cap), kBlock, s, (const int64_t *)out_rowptr, B_cap, E_cap, local, e_pos,
               out_col, out_eid);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_sample_hop(const int64_t *rowptr, const int64_t *col, const int64_t *seeds,
                              const int64_t *n_seeds_dev, int64_t B_cap, int64_t num_nodes, int64_t fanout,
                              int64_t E_cap, int64_t S_cap, int64_t *rng_state, int64_t *first_pos, int64_t *out_rowptr,
                              int32_t *out_col, int64_t *out_eid, int64_t *out_nid, int64_t *out_counts,
                              void *workspace, size_t workspace_bytes, void *stream) {
  return ggl_sample_hop_ex(rowptr, col, seeds, n_seeds_dev, B_cap, num_nodes, fanout, E_cap, S_cap, rng_state, first_pos,
                           out_rowptr, out_col, out_eid, out_nid, out_counts, workspace, workspace_bytes, nullptr, stream);
}

extern "C" size_t ggl_block_transpose_workspace_bytes(int64_t E_cap, int64_t N_src_cap) {
  if (E_cap < 0 || N_src_cap < 0) return 0;
  size_t b = 3 * up256((size_t)(E_cap > 0 ? E_cap : 1) * 4);  // keys in / out, vals in
#ifndef GGL_EMULATE
  b += up256(sortpairs_temp_bytes(E_cap > 0 ? E_cap : 1, bits_for(N_src_cap)));
#endif
  return b + 256;
}

// rowptrT[N_src_cap + 1] / dstT[E_cap]: for source row j the destination rows of its block edges, ascending —
// the CSC of a sampled block, built without reading anything back (graph-capture safe).
extern "C" int ggl_block_transpose(const int64_t *rowptr, const int32_t *col, int64_t N_dst, int64_t N_src_cap,
                                   int64_t E_cap, int64_t *rowptrT, int32_t *dstT, void *workspace,
                                   size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(N_dst >= 0 && N_src_cap >= 0 && E_cap >= 0 && E_cap < ((int64_t)1 << 31) &&
                  N_src_cap < ((int64_t)1 << 31) - 1, GGL_EINVAL, "bad sizes");
  GGL_REQUIRE(rowptr && rowptrT, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_block_transpose_workspace_bytes(E_cap, N_src_cap), GGL_EWORKSPACE,
              "block_transpose workspace too small");
  hipStream_t s = as_stream(stream);
  char *ws = static_cast<char *>(workspace);
  const size_t seg = up256((size_t)(E_cap > 0 ? E_cap : 1) * 4);
  uint32_t *keys_in = reinterpret_cast<uint32_t *>(ws), *keys_out = reinterpret_cast<uint32_t *>(ws + seg);
  int32_t *vals_in = reinterpret_cast<int32_t *>(ws + 2 * seg);
  if (E_cap > 0) {
    GGL_REQUIRE(col && dstT, GGL_EINVAL, "NULL pointer");
    GGL_LAUNCH((block_keys_kernel), grid_for(E_cap), kBlock, s, rowptr, N_dst, col, E_cap, N_src_cap, keys_in, vals_in);
    GGL_LAUNCH_CHECK();
#ifndef GGL_EMULATE
    const int bits = bits_for(N_src_cap);
    size_t tmp = sortpairs_temp_bytes(E_cap, bits);
    GGL_HIP_CHECK(rocprim::radix_sort_pairs(ws + 3 * seg, tmp, (const uint32_t *)keys_in, keys_out,
                                            (const int32_t *)vals_in, dstT, (size_t)E_cap, 0u, (unsigned)bits, s));
#else
    std::vector<int32_t> order((size_t)E_cap);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return keys_in[a] < keys_in[b]; });
    for (int64_t i = 0; i < E_cap; ++i) {
      keys_out[i] = keys_in[order[(size_t)i]];
      dstT[i] = vals_in[order[(size_t)i]];
    }
#endif
  }
  GGL_LAUNCH((block_rowptrT_kernel), grid_for(N_src_cap + 1), kBlock, s, (const uint32_t *)keys_out, E_cap, N_src_cap,
             rowptrT);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}
