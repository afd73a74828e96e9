// gammagl_amd/csrc/gat.hip — fused GAT edge-softmax + weighted aggregate.
//
// Replaces (a) the external dgNN GATConvFuse CUDA kernel that FusedGATConv calls
// (layers/conv/fusedgat_conv.py:70-71,121 — not in the reference tree, parity unpinned) and (b) the
// unfused chain GATConv.forward runs today (layers/conv/gat_conv.py:103-112 + utils/softmax.py:29-35):
// 2 gathers [E,H,C] + concat + reduce -> leaky_relu -> segment_max -> gather -> exp -> segment_sum ->
// gather -> divide -> gather [E,H,C] * alpha -> segment_sum, i.e. three segment passes and five
// [E,.] intermediates in HBM.
//
// Forward (ONE launch + a tiny combine for hub rows): a lane group owns one destination row, each lane
// VEC channels of one head, and walks the row ONCE with an online softmax (gat_online below): running
// max, denominator and weighted sum, rescaled when the max moves; out = acc / (den + 1e-16).  (The first
// version walked the row three times as the in-tree math does — max, denominator, weighted sum — and
// was latency-bound on the two extra index/el walks: 9.3 ms on the Reddit-sized graph.)
// Lanes of the same head recompute the (cheap) scalar softmax terms redundantly instead of
// exchanging them: no LDS, no shuffles, no atomics.
// Rows longer than plan->chunk (a Reddit-sized R-MAT graph has a 109 110-edge hub: 100 ms on one lane
// group) are cut into chunks reduced by whole wavefronts in the leading blocks of the same launch with
// a chunk-local max (m_c, d_c = sum exp(s - m_c), acc_c = sum exp(s - m_c) x); gat_long_final_kernel
// merges them in chunk order: m = max m_c, d = sum d_c e^{m_c - m}, out = sum acc_c e^{m_c - m} / (d + 1e-16).
//
// Backward: two walks, one per side.
//   gat_bwd_dst_kernel (forward plan, lane = head of a destination row or of a hub chunk): row dot
//     <g_i, out_i>, alpha, de = alpha (<g_i, x_j> - dot) LeakyReLU'(.) -> alpha[E,H], de[E,H], and
//     ger[i,h] = sum_p de in the same walk;
//   gat_bwd_src_kernel (transposed plan): gx[j,h,:] = sum alpha g[dst,h,:] and gel[j,h] = sum de, reading
//     alpha / de through posT.
// (The first version was edge-parallel on the destination side — a row-dot kernel, one thread per
// (position, head), then ggl_segment_sum for ger — and ran bspmm + segment_sum on the source side.)
// Roofline: HBM; algorithmic bytes per edge = 4*H*C (feature row) + 4 (col) + 4*H (el row).
#include "gat_common.hpp"

namespace ggl {

// One walk over positions [beg, end) of a destination row for head h, channels [kk, kk+VEC):
//   m   = max_p s_p,   s_p = LeakyReLU(el[col[p],h] + er_i)
//   den = sum_p exp(s_p - m)
//   acc = sum_p exp(s_p - m) * x[col[p], kk:kk+VEC]
// computed online: the running (den, acc) are rescaled by exp(m_old - m_new) whenever a new maximum
// appears (O(log len) times on average), so every feature row, el value and column index is read exactly
// once.  The in-tree chain (softmax.py:29-35) makes three passes (max, sum, weighted sum); the one-walk
// form differs from it only in rounding (a few ulp per rescale; the parity bar for float reductions is
// 1e-5 relative and is tested against the oracle's three-pass restatement).  Four feature rows in flight.
// DROP: attention dropout — the softmax statistics (m, den) see every edge, the weighted sum only the
// kept ones, scaled by 1/(1-p): out = sum_p keep_p alpha_p x_p / (1-p), alpha = softmax over ALL edges.
// Random word of (position p, head h): drop_word(p, h) above — a 15-instruction counter-based mix, so a draw per
// edge is affordable in every walk (with Philox4x32-10 a draw per edge doubled the forward: 4.1 -> 9.4 ms, and the
// walks were aligned to multiples of 4 to share one 4-word block; the alignment is kept for the 16-byte index loads).
template <int VEC, bool DROP>
__device__ __forceinline__ void gat_online(const int32_t *__restrict__ col, const float *__restrict__ el,
                                           const float *__restrict__ x, float er_i, float slope, int64_t H,
                                           int64_t K, int64_t h, int64_t kk, int64_t beg, int64_t end,
                                           const GatDims &d, const int64_t *__restrict__ rng,
                                           float &m, float &den, float (&acc)[VEC]) {
  m = -FLT_MAX;  // unsorted_segment_max: lowest() fill, strict <
  den = 0.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  auto absorb = [&](uint32_t word, float s, const float (&v)[VEC]) {
    if (m < s) {  // new maximum: bring the running sums to the new reference point
      const float sc = GGL_EXPF(__fadd_rn(m, -s));  // m = -FLT_MAX on the first element: exp(-inf) = 0
      den = __fmul_rn(den, sc);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = __fmul_rn(acc[i], sc);
      m = s;
    }
    const float e = GGL_EXPF(__fadd_rn(s, -m));
    den = __fadd_rn(den, e);
    float ek = e;
    if (DROP) ek = (word >= d.drop_thresh) ? __fmul_rn(e, d.drop_scale) : 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = __fadd_rn(acc[i], __fmul_rn(v[i], ek));
  };
  auto single = [&](int64_t q) {
    const int64_t c0 = col[q];
    float v0[VEC];
    F32V<VEC>::load(x + c0 * K + kk, v0);
    absorb(DROP ? drop_word(q, H, h, offset, seed) : 0u, lrelu(__fadd_rn(el[c0 * H + h], er_i), slope), v0);
  };
  int64_t p = beg;
  if (DROP) {  // walk up to a multiple of 4 so that each unrolled group shares one draw
    for (; p < end && (p & 3) != 0; ++p) single(p);
  }
  for (; p + 4 <= end; p += 4) {
    int64_t c[4];
    float v[4][VEC], s[4];
    U4 rw{0u, 0u, 0u, 0u};
    if (DROP) rw = drop_words4(p >> 2, H, h, offset, seed);
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = col[p + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) F32V<VEC>::load(x + c[u] * K + kk, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = lrelu(__fadd_rn(el[c[u] * H + h], er_i), slope);
#pragma unroll
    for (int u = 0; u < 4; ++u) absorb(pick_word(rw, u), s[u], v[u]);
  }
  for (; p < end; ++p) single(p);
}

template <int VEC, bool DROP>
__global__ __launch_bounds__(kBlock) void gat_fwd_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ er,
    const float *__restrict__ x, float *__restrict__ y, float *__restrict__ rowmax,
    float *__restrict__ rowden, float *__restrict__ pacc, float *__restrict__ pm,
    float *__restrict__ pd, const int64_t *__restrict__ rng, const GatDims d) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t H = d.H, K = d.K;
  if (block_id() < d.chunk_blocks) {  // one wavefront per chunk of a long row
    const int64_t cid = block_id() * kWavesPerBlock + wave;
    if (cid >= d.n_chunks) return;
    int64_t lo = 0, hi = d.n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= cid) lo = mid; else hi = mid - 1;
    }
    const int64_t row = long_rows[lo];
    const int64_t beg = rowptr[row] + (cid - chunk_ptr[lo]) * d.chunk;
    const int64_t rend = rowptr[row + 1];
    const int64_t end = (beg + d.chunk < rend) ? beg + d.chunk : rend;
    for (int64_t kk = (int64_t)lane * VEC; kk < K; kk += (int64_t)kWave * VEC) {
      const int64_t h = kk / d.C;
      const float er_i = er[row * H + h];
      float m, dsum, acc[VEC];
      gat_online<VEC, DROP>(col, el, x, er_i, d.slope, H, K, h, kk, beg, end, d, rng, m, dsum, acc);
      F32V<VEC>::store(pacc + cid * K + kk, acc);
      if (kk == h * d.C) {
        pm[cid * H + h] = m;
        pd[cid * H + h] = dsum;
      }
    }
    return;
  }
  const int L = 1 << d.logL;
  const int64_t slot = ((block_id() - d.chunk_blocks) * kWavesPerBlock + wave) * (kWave >> d.logL) +
                       (lane >> d.logL);
  if (slot >= d.N) return;
  const int64_t row = row_order ? (int64_t)row_order[slot] : slot;
  const int li = lane & (L - 1);
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  if (end - beg > d.chunk) return;
  for (int64_t kk = (int64_t)li * VEC; kk < K; kk += (int64_t)L * VEC) {
    const int64_t h = kk / d.C;
    const float er_i = er[row * H + h];
    float m, dsum, acc[VEC];
    gat_online<VEC, DROP>(col, el, x, er_i, d.slope, H, K, h, kk, beg, end, d, rng, m, dsum, acc);
    const float inv = __fadd_rn(dsum, 1e-16f);  // softmax.py:35: exp / (sum + 1e-16)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = __fdiv_rn(acc[i], inv);
    F32V<VEC>::store(y + row * K + kk, acc);
    if (kk == h * d.C) {  // first lane of the head records the softmax statistics
      rowmax[row * H + h] = m;
      rowden[row * H + h] = dsum;
    }
  }
}

__global__ __launch_bounds__(kBlock) void gat_long_final_kernel(
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr,
    const float *__restrict__ pacc, const float *__restrict__ pm, const float *__restrict__ pd,
    float *__restrict__ y, float *__restrict__ rowmax, float *__restrict__ rowden, const GatDims d) {
  const int64_t j = block_id();
  if (j >= d.n_long) return;
  const int64_t row = long_rows[j];
  const int64_t c0 = chunk_ptr[j], c1 = chunk_ptr[j + 1];
  for (int64_t k = threadIdx.x; k < d.K; k += kBlock) {
    const int64_t h = k / d.C;
    float m = -FLT_MAX;
    for (int64_t c = c0; c < c1; ++c)
      if (m < pm[c * d.H + h]) m = pm[c * d.H + h];
    float den = 0.0f, acc = 0.0f;
    for (int64_t c = c0; c < c1; ++c) {
      const float sc = GGL_EXPF(__fadd_rn(pm[c * d.H + h], -m));
      den = __fadd_rn(den, __fmul_rn(pd[c * d.H + h], sc));
      acc = __fadd_rn(acc, __fmul_rn(pacc[c * d.K + k], sc));
    }
    y[row * d.K + k] = __fdiv_rn(acc, __fadd_rn(den, 1e-16f));
    if (k == h * d.C) {
      rowmax[row * d.H + h] = m;
      rowden[row * d.H + h] = den;
    }
  }
}

// Destination-major half of the backward in ONE walk of the forward plan.  A work item is a short row or
// one chunk of a long row; a group of 2^logL >= H lanes owns it, lane = head.  Per head the lane computes
// dot_i = <g_i, out_i> once, then walks the item's positions in order:
//   alpha_p = exp(s_p - m_i) / (den_i + 1e-16),  dalpha_p = <g_i[h,:], x_j[h,:]>,
//   de_p = alpha_p (dalpha_p - dot_i) LeakyReLU'(raw_p)          -> alpha[E,H], de[E,H] (sorted positions)
//   ger[i,h] = sum_p de_p  (in position order; chunk partials are combined in chunk order afterwards)
// CREG > 0: C is known at compile time and g_i[h,:] lives in registers; 0 = any C, g_i re-read (cached).
// Replaces three launches (row dots, an edge-parallel alpha/de kernel, segment_sum(de) for ger: 0.3 + 5.7
// + 2.7 ms on the Reddit-sized graph) with the same rounded operations in the same order.
template <int VEC, int CREG>
__global__ __launch_bounds__(kBlock) void gat_bwd_dst_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ er,
    const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ out,
    const float *__restrict__ rowmax, const float *__restrict__ rowden, float *__restrict__ alpha,
    float *__restrict__ de, float *__restrict__ ger, float *__restrict__ pger,
    const int64_t *__restrict__ rng, const GatDims d) {
  const int64_t H = d.H, K = d.K;
  const int64_t C = CREG > 0 ? (int64_t)CREG : d.C;
  const uint64_t seed = d.drop_thresh ? (uint64_t)rng[0] : 0, offset = d.drop_thresh ? (uint64_t)rng[1] : 0;
  const int LG = 1 << d.logL;
  const int64_t item = (block_id() * kBlock + threadIdx.x) >> d.logL;
  const int li = threadIdx.x & (LG - 1);
  if (item >= d.n_chunks + d.N) return;
  const bool is_chunk = item < d.n_chunks;
  int64_t row, beg, end;
  if (is_chunk) {  // chunks carry the lowest item ids: the hub work is dispatched first
    int64_t lo = 0, hi = d.n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= item) lo = mid; else hi = mid - 1;
    }
    row = long_rows[lo];
    beg = rowptr[row] + (item - chunk_ptr[lo]) * d.chunk;
    const int64_t rend = rowptr[row + 1];
    end = (beg + d.chunk < rend) ? beg + d.chunk : rend;
  } else {
    const int64_t slot = item - d.n_chunks;
    row = row_order ? (int64_t)row_order[slot] : slot;
    beg = rowptr[row];
    end = rowptr[row + 1];
    if (end - beg > d.chunk) return;  // long row: its chunks are separate items
  }
  for (int64_t h = li; h < H; h += LG) {
    const float *__restrict__ gi = g + row * K + h * C;
    const float *__restrict__ oi = out + row * K + h * C;
    float gr[CREG > 0 ? CREG : 1];
    float dot = 0.0f;
    if (CREG > 0) {
#pragma unroll
      for (int c = 0; c < (CREG > 0 ? CREG : 1); ++c) {
        gr[c] = gi[c];
        dot = __fadd_rn(dot, __fmul_rn(gr[c], oi[c]));
      }
    } else {
      for (int64_t c = 0; c < C; ++c) dot = __fadd_rn(dot, __fmul_rn(gi[c], oi[c]));
    }
    const float er_i = er[row * H + h];
    const float m = rowmax[row * H + h];
    const float inv = __fadd_rn(rowden[row * H + h], 1e-16f);
    float gsum = 0.0f;
    auto edge = [&](int64_t p, uint32_t word, float elv, const float *__restrict__ xj) {
      const float raw = __fadd_rn(elv, er_i);
      const float al = __fdiv_rn(GGL_EXPF(__fadd_rn(lrelu(raw, d.slope), -m)), inv);
      float da = 0.0f;
      if (CREG > 0) {
#pragma unroll
        for (int c = 0; c < (CREG > 0 ? CREG : 1); c += VEC) {
          float xv[VEC];
          F32V<VEC>::load(xj + c, xv);
#pragma unroll
          for (int q = 0; q < VEC; ++q) da = __fadd_rn(da, __fmul_rn(gr[(c + q) % (CREG > 0 ? CREG : 1)], xv[q]));
        }
      } else {
        for (int64_t c = 0; c < C; c += VEC) {
          float gv[VEC], xv[VEC];
          F32V<VEC>::load(gi + c, gv);
          F32V<VEC>::load(xj + c, xv);
#pragma unroll
          for (int q = 0; q < VEC; ++q) da = __fadd_rn(da, __fmul_rn(gv[q], xv[q]));
        }
      }
      float alk = al;  // the weight the edge carried forward: alpha, or keep * alpha / (1 - p)
      if (d.drop_thresh) {  // same draw as the forward (same seed, offset, index)
        const bool keep = word >= d.drop_thresh;
        alk = keep ? __fmul_rn(al, d.drop_scale) : 0.0f;
        da = keep ? __fmul_rn(da, d.drop_scale) : 0.0f;  // d out / d alpha_p = keep/(1-p) <g_i, x_j>
      }
      const float ds = __fmul_rn(al, __fadd_rn(da, -dot));
      const float dv = raw > 0.0f ? ds : __fmul_rn(ds, d.slope);
      alpha[(p * H + h) * d.es] = alk;
      de[(p * H + h) * d.es] = dv;
      gsum = __fadd_rn(gsum, dv);
    };
    auto single = [&](int64_t q) {
      const int64_t s0 = col[q];
      edge(q, d.drop_thresh ? drop_word(q, H, h, offset, seed) : 0u, el[s0 * H + h], x + s0 * K + h * C);
    };
    int64_t p = beg;
    if (d.drop_thresh) {
      for (; p < end && (p & 3) != 0; ++p) single(p);
    }
    for (; p + 4 <= end; p += 4) {
      int64_t sj[4];
      float ev[4];
      U4 rw{0u, 0u, 0u, 0u};
      if (d.drop_thresh) rw = drop_words4(p >> 2, H, h, offset, seed);
#pragma unroll
      for (int u = 0; u < 4; ++u) sj[u] = col[p + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) ev[u] = el[sj[u] * H + h];
#pragma unroll
      for (int u = 0; u < 4; ++u) edge(p + u, pick_word(rw, u), ev[u], x + sj[u] * K + h * C);
    }
    for (; p < end; ++p) single(p);
    if (is_chunk) pger[item * H + h] = gsum;
    else ger[row * H + h] = gsum;
  }
}

// ---- wide heads (C > 16), GPU build only.  The lane-per-head walk above reads each lane's C-float strip
// with C/4 strided 16-byte loads and collapsed on the Reddit GAT's last layer (8 heads x 41 classes: 222 ms at
// C = 40, 870 ms at C = 41).  Here a GROUP of 2^LOGG lanes owns the row (or hub chunk) and the lanes split
// the CHANNELS of every head: lane s holds g_i[h, 4s..4s+3] for all HH heads in registers, reads the same
// slice of x_j per head (a head's strip is one coalesced read, the HH reads of an edge cover its whole
// contiguous row), and the per-edge dots <g_i[h,:], x_j[h,:]> are reduced across the group with a butterfly
// of wave shuffles — the one place this library reduces across lanes: GAT gradients are held to 1e-5
// relative, not bit-exact, so the association may change.  After the butterflies every lane holds every
// dot; lane h finishes head h (alpha, de, running ger sum in position order).  The host emulation build
// cannot shuffle between its sequentially executed lanes and keeps using gat_bwd_dst_kernel for every C.
#ifndef GGL_EMULATE
template <int LOGG> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = (1 << LOGG) >> 1; o > 0; o >>= 1) v = __fadd_rn(v, __shfl_xor(v, o, 64));
  return v;
}

template <int VEC, int LOGG, int HH>
__global__ __launch_bounds__(kBlock) void gat_bwd_dst_wide_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ er,
    const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ out,
    const float *__restrict__ rowmax, const float *__restrict__ rowden, float *__restrict__ alpha,
    float *__restrict__ de, float *__restrict__ ger, float *__restrict__ pger,
    const int64_t *__restrict__ rng, const GatDims d) {
  constexpr int G = 1 << LOGG;
  static_assert(G >= HH, "lane h finishes head h");
  const int64_t H = d.H, C = d.C, K = d.K;  // H <= HH: HH is the compile-time bound of the head loops
  const int64_t item = thread_id() >> LOGG;
  const int sub = (int)(threadIdx.x & (G - 1));
  if (item >= d.n_chunks + d.N) return;  // whole groups leave together: the shuffles below stay in-group
  const bool is_chunk = item < d.n_chunks;
  int64_t row, beg, end;
  if (is_chunk) {
    int64_t lo = 0, hi = d.n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= item) lo = mid; else hi = mid - 1;
    }
    row = long_rows[lo];
    beg = rowptr[row] + (item - chunk_ptr[lo]) * d.chunk;
    const int64_t rend = rowptr[row + 1];
    end = (beg + d.chunk < rend) ? beg + d.chunk : rend;
  } else {
    const int64_t slot = item - d.n_chunks;
    row = row_order ? (int64_t)row_order[slot] : slot;
    beg = rowptr[row];
    end = rowptr[row + 1];
    if (end - beg > d.chunk) return;
  }
  const int64_t c0 = (int64_t)sub * VEC;
  const bool act = c0 < C;  // C % VEC == 0 on this path
  float gr[HH][VEC], dots[HH];
#pragma unroll
  for (int h = 0; h < HH; ++h) {
    float ov[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) { gr[h][q] = 0.0f; ov[q] = 0.0f; }
    if (act && h < H) {
      F32V<VEC>::load(g + row * K + h * C + c0, gr[h]);
      F32V<VEC>::load(out + row * K + h * C + c0, ov);
    }
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) t = __fadd_rn(t, __fmul_rn(gr[h][q], ov[q]));
    dots[h] = group_sum<LOGG>(t);
  }
  // lane h keeps the row constants of head h
  float my_dot = 0.0f, my_er = 0.0f, my_m = 0.0f, my_inv = 1.0f;
#pragma unroll
  for (int h = 0; h < HH; ++h)
    if (sub == h) my_dot = dots[h];
  if (sub < H) {
    my_er = er[row * H + sub];
    my_m = rowmax[row * H + sub];
    my_inv = __fadd_rn(rowden[row * H + sub], 1e-16f);
  }
  const uint64_t seed = d.drop_thresh ? (uint64_t)rng[0] : 0, offset = d.drop_thresh ? (uint64_t)rng[1] : 0;
  float gsum = 0.0f;
  auto one_edge = [&](int64_t p, int64_t j, const float (&xv)[HH][VEC]) {
    float part[HH];
#pragma unroll
    for (int h = 0; h < HH; ++h) {
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < VEC; ++q) t = __fadd_rn(t, __fmul_rn(gr[h][q], xv[h][q]));
      part[h] = group_sum<LOGG>(t);
    }
    if (sub < H) {
      float da = 0.0f;
#pragma unroll
      for (int h = 0; h < HH; ++h)
        if (sub == h) da = part[h];
      const float raw = __fadd_rn(el[j * H + sub], my_er);
      const float al = __fdiv_rn(GGL_EXPF(__fadd_rn(lrelu(raw, d.slope), -my_m)), my_inv);
      float alk = al;
      if (d.drop_thresh) {
        const bool keep = drop_word(p, H, sub, offset, seed) >= d.drop_thresh;
        alk = keep ? __fmul_rn(al, d.drop_scale) : 0.0f;
        da = keep ? __fmul_rn(da, d.drop_scale) : 0.0f;
      }
      const float ds = __fmul_rn(al, __fadd_rn(da, -my_dot));
      const float dv = raw > 0.0f ? ds : __fmul_rn(ds, d.slope);
      alpha[(p * H + sub) * d.es] = alk;
      de[(p * H + sub) * d.es] = dv;
      gsum = __fadd_rn(gsum, dv);
    }
  };
  int64_t p = beg;
  for (; p + 2 <= end; p += 2) {  // two feature rows (2 x HH slices per lane) in flight
    const int64_t j0 = col[p], j1 = col[p + 1];
    float x0[HH][VEC], x1[HH][VEC];
#pragma unroll
    for (int h = 0; h < HH; ++h) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) { x0[h][q] = 0.0f; x1[h][q] = 0.0f; }
      if (act && h < H) {
        F32V<VEC>::load(x + j0 * K + h * C + c0, x0[h]);
        F32V<VEC>::load(x + j1 * K + h * C + c0, x1[h]);
      }
    }
    one_edge(p, j0, x0);
    one_edge(p + 1, j1, x1);
  }
  for (; p < end; ++p) {
    const int64_t j0 = col[p];
    float x0[HH][VEC];
#pragma unroll
    for (int h = 0; h < HH; ++h) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) x0[h][q] = 0.0f;
      if (act && h < H) F32V<VEC>::load(x + j0 * K + h * C + c0, x0[h]);
    }
    one_edge(p, j0, x0);
  }
  if (sub < H) {
    if (is_chunk) pger[item * H + sub] = gsum;
    else ger[row * H + sub] = gsum;
  }
}
#endif  // !GGL_EMULATE

__global__ __launch_bounds__(kBlock) void gat_bwd_dst_final_kernel(const int32_t *__restrict__ long_rows,
                                                                   const int64_t *__restrict__ chunk_ptr,
                                                                   const float *__restrict__ pger,
                                                                   float *__restrict__ ger, int64_t n_long,
                                                                   int64_t H) {
  const int64_t stride = grid_threads();
  for (int64_t t = thread_id(); t < n_long * H; t += stride) {
    const int64_t j = t / H, h = t - j * H;
    float a = 0.0f;
    for (int64_t c = chunk_ptr[j]; c < chunk_ptr[j + 1]; ++c) a = __fadd_rn(a, pger[c * H + h]);
    ger[(int64_t)long_rows[j] * H + h] = a;
  }
}

// Source-major half of the backward in ONE walk of the transposed plan (rows = source nodes j):
//   gx[j,h,:] = sum_q alpha[posT[q],h] * g[colT[q],h,:]        gel[j,h] = sum_q de[posT[q],h]
// posT maps a transposed position to the forward (destination-sorted) position alpha / de were written
// at.  Same lane layout as the forward; the first lane of each head also carries the gel sum.  Same
// rounded operations in the same order as the bspmm + segment_sum pair this replaces (two walks, two
// gathers of posT: 6.0 + 3.7 ms on the Reddit-sized graph), so the results are bit-identical to it.
template <int VEC>
__device__ __forceinline__ void gat_src_walk(const int32_t *__restrict__ colT, const int32_t *__restrict__ posT,
                                             const float *__restrict__ alpha, const float *__restrict__ de,
                                             const float *__restrict__ g, int64_t H, int64_t K, int64_t h,
                                             int64_t es, int64_t kk, bool lead, int64_t beg, int64_t end,
                                             float (&acc)[VEC], float &gl) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
  gl = 0.0f;
  int64_t q = beg;
  for (; q + 4 <= end; q += 4) {
    int64_t r[4], e[4];
    float v[4][VEC], a[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { r[u] = colT[q + u]; e[u] = posT[q + u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) F32V<VEC>::load(g + r[u] * K + kk, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = alpha[(e[u] * H + h) * es];
      dv[u] = lead ? de[(e[u] * H + h) * es] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = __fadd_rn(acc[i], __fmul_rn(a[u], v[u][i]));
      gl = __fadd_rn(gl, dv[u]);
    }
  }
  for (; q < end; ++q) {
    const int64_t r0 = colT[q], e0 = posT[q];
    float v0[VEC];
    F32V<VEC>::load(g + r0 * K + kk, v0);
    const float a0 = alpha[(e0 * H + h) * es];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = __fadd_rn(acc[i], __fmul_rn(a0, v0[i]));
    if (lead) gl = __fadd_rn(gl, de[(e0 * H + h) * es]);
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gat_bwd_src_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colT, const int32_t *__restrict__ posT,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ alpha, const float *__restrict__ de,
    const float *__restrict__ g, float *__restrict__ gx, float *__restrict__ gel, float *__restrict__ pacc,
    float *__restrict__ pgel, const GatDims d) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t H = d.H, K = d.K;
  if (block_id() < d.chunk_blocks) {  // one wavefront per chunk of a long row
    const int64_t cid = block_id() * kWavesPerBlock + wave;
    if (cid >= d.n_chunks) return;
    int64_t lo = 0, hi = d.n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= cid) lo = mid; else hi = mid - 1;
    }
    const int64_t row = long_rows[lo];
    const int64_t beg = rowptr[row] + (cid - chunk_ptr[lo]) * d.chunk;
    const int64_t rend = rowptr[row + 1];
    const int64_t end = (beg + d.chunk < rend) ? beg + d.chunk : rend;
    for (int64_t kk = (int64_t)lane * VEC; kk < K; kk += (int64_t)kWave * VEC) {
      const int64_t h = kk / d.C;
      const bool lead = (kk == h * d.C);
      float acc[VEC], gl;
      gat_src_walk<VEC>(colT, posT, alpha, de, g, H, K, h, d.es, kk, lead, beg, end, acc, gl);
      F32V<VEC>::store(pacc + cid * K + kk, acc);
      if (lead) pgel[cid * H + h] = gl;
    }
    return;
  }
  const int L = 1 << d.logL;
  const int64_t slot = ((block_id() - d.chunk_blocks) * kWavesPerBlock + wave) * (kWave >> d.logL) +
                       (lane >> d.logL);
  if (slot >= d.N) return;
  const int64_t row = row_order ? (int64_t)row_order[slot] : slot;
  const int li = lane & (L - 1);
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  if (end - beg > d.chunk) return;
  for (int64_t kk = (int64_t)li * VEC; kk < K; kk += (int64_t)L * VEC) {
    const int64_t h = kk / d.C;
    const bool lead = (kk == h * d.C);
    float acc[VEC], gl;
    gat_src_walk<VEC>(colT, posT, alpha, de, g, H, K, h, d.es, kk, lead, beg, end, acc, gl);
    F32V<VEC>::store(gx + row * K + kk, acc);
    if (lead) gel[row * H + h] = gl;
  }
}

// long source rows: partial sums combined in chunk order (deterministic)
__global__ __launch_bounds__(kBlock) void gat_bwd_src_final_kernel(
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr,
    const float *__restrict__ pacc, const float *__restrict__ pgel, float *__restrict__ gx,
    float *__restrict__ gel, const GatDims d) {
  const int64_t j = block_id();
  if (j >= d.n_long) return;
  const int64_t row = long_rows[j];
  const int64_t c0 = chunk_ptr[j], c1 = chunk_ptr[j + 1];
  for (int64_t k = threadIdx.x; k < d.K + d.H; k += kBlock) {
    float acc = 0.0f;
    if (k < d.K) {
      for (int64_t c = c0; c < c1; ++c) acc = __fadd_rn(acc, pacc[c * d.K + k]);
      gx[row * d.K + k] = acc;
    } else {
      const int64_t h = k - d.K;
      for (int64_t c = c0; c < c1; ++c) acc = __fadd_rn(acc, pgel[c * d.H + h]);
      gel[row * d.H + h] = acc;
    }
  }
}


}  // namespace ggl

using namespace ggl;

extern "C" size_t ggl_gat_partial_bytes(int64_t n_chunks, int64_t H, int64_t C) {
  if (n_chunks <= 0) return 0;
  return (size_t)n_chunks * (size_t)(H * C + 2 * H) * sizeof(float) + 64;
}


extern "C" int ggl_gat_fused_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el,
                                 const float *er, const float *x, float slope, int64_t H, int64_t C,
                                 float p_drop, int64_t *rng_state, float *out, float *rowmax,
                                 float *rowden, void *stream) {
  GGL_REQUIRE(plan && plan->rowptr, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(H > 0 && C > 0 && plan->chunk > 0, GGL_EINVAL, "H, C and chunk must be positive");
  const int64_t N = plan->N;
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(er && out && rowmax && rowden, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((col && el && x) || plan->E == 0, GGL_EINVAL, "NULL pointer");
  GatDims d{};
  d.slope = slope; d.N = N; d.H = H; d.C = C; d.K = H * C; d.E = plan->E;
  d.chunk = plan->chunk; d.n_long = plan->n_long; d.n_chunks = plan->n_chunks;
  int rcd = set_dropout(d, p_drop, rng_state);
  if (rcd) return rcd;
  float *pacc = nullptr, *pm = nullptr, *pd = nullptr;
  if (plan->n_long > 0) {
    GGL_REQUIRE(plan->long_rows && plan->chunk_ptr && plan->partial, GGL_EWORKSPACE,
                "plan has long rows but long_rows/chunk_ptr/partial is NULL");
    pacc = static_cast<float *>(plan->partial);
    pm = pacc + plan->n_chunks * d.K;
    pd = pm + plan->n_chunks * H;
    d.chunk_blocks = ceil_div(plan->n_chunks, kWavesPerBlock);
  }
  const bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(pacc) & 15u) == 0) && !options().force_generic;
  const int vec = vec4 ? 4 : 1;
  d.logL = pow2_log2(ceil_div(d.K, vec));
  d.nblocks = ceil_div(N, (int64_t)kWavesPerBlock * (kWave >> d.logL));
  const int64_t grid = d.chunk_blocks + d.nblocks;
  GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  const int32_t *order = options().row_order ? plan->row_order : nullptr;
  hipStream_t s = as_stream(stream);
#define GGL_GAT_FWD(V, DR)                                                                              \
  GGL_LAUNCH((gat_fwd_kernel<V, DR>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,       \
             plan->chunk_ptr, el, er, x, out, rowmax, rowden, pacc, pm, pd, (const int64_t *)rng_state, d)
  if (vec4 && d.drop_thresh) GGL_GAT_FWD(4, true);
  else if (vec4) GGL_GAT_FWD(4, false);
  else if (d.drop_thresh) GGL_GAT_FWD(1, true);
  else GGL_GAT_FWD(1, false);
#undef GGL_GAT_FWD
  GGL_LAUNCH_CHECK();
  if (plan->n_long > 0) {
    GGL_LAUNCH((gat_long_final_kernel), plan->n_long, kBlock, s, plan->long_rows, plan->chunk_ptr,
               (const float *)pacc, (const float *)pm, (const float *)pd, out, rowmax, rowden, d);
    GGL_LAUNCH_CHECK();
  }
  if (d.drop_thresh) return rng_advance(rng_state, stream);  // the next call draws a new mask
  return GGL_OK;
}

extern "C" int ggl_gat_fused_bwd_dst(const ggl_segplan_t *plan, const int32_t *col,
                                     const int32_t *rowidx, const float *el, const float *er,
                                     const float *x, const float *g, const float *out,
                                     const float *rowmax, const float *rowden, float slope, int64_t H,
                                     int64_t C, float p_drop, const int64_t *rng_used, float *alpha,
                                     float *de, float *ger, float *dot_ws, void *stream) {
  (void)rowidx; (void)dot_ws;  // needed by the first (edge-parallel) version; accepted, unused
  GGL_REQUIRE(plan && plan->rowptr, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(H > 0 && C > 0 && plan->chunk > 0, GGL_EINVAL, "H, C and chunk must be positive");
  const int64_t N = plan->N, E = plan->E;
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(er && g && out && rowmax && rowden && ger, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((col && el && x && alpha && de) || E == 0, GGL_EINVAL, "NULL pointer");
  GatDims d{};
  d.slope = slope; d.N = N; d.H = H; d.C = C; d.K = H * C; d.E = E;
  d.es = (de == alpha + 1) ? 2 : 1;  // de == alpha + 1: one interleaved [E,H,2] buffer
  d.chunk = plan->chunk; d.n_long = plan->n_long; d.n_chunks = plan->n_long > 0 ? plan->n_chunks : 0;
  int rcd = set_dropout(d, p_drop, rng_used);
  if (rcd) return rcd;
  float *pger = nullptr;
  if (plan->n_long > 0) {
    GGL_REQUIRE(plan->long_rows && plan->chunk_ptr && plan->partial, GGL_EWORKSPACE,
                "plan has long rows but long_rows/chunk_ptr/partial is NULL");
    pger = static_cast<float *>(plan->partial);
  }
  d.logL = pow2_log2(H);  // lanes per work item: the next power of two >= H, at most 64
  const int64_t items = d.n_chunks + N;
  const int64_t grid = ceil_div(items << d.logL, (int64_t)kBlock);
  GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  const int32_t *order = options().row_order ? plan->row_order : nullptr;
  hipStream_t s = as_stream(stream);
  const bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) && !options().force_generic;
#ifndef GGL_EMULATE
  // wide heads: lanes split the channels, shuffle-reduced dots (gat_bwd_dst_wide_kernel); needs a head count
  // the kernel is instantiated for and a head that fits one group (C <= 64 * vec)
  {
    const int vec = vec4 ? 4 : 1;
    const bool h_ok = H <= 16;  // instantiated head-loop bounds: 1, 2, 4, 8, 16 (the next one >= H is used)
    if (C > 16 && h_ok && C <= 64 * vec && !options().force_generic) {
      int logg = pow2_log2(ceil_div(C, vec));
      if (logg < 4) logg = 4;  // >= 16 lanes: covers H <= 16 finishing lanes
      const int64_t wgrid = ceil_div(items << logg, (int64_t)kBlock);
#define GGL_GAT_WIDE(V, LG, HH)                                                                          \
  GGL_LAUNCH((gat_bwd_dst_wide_kernel<V, LG, HH>), wgrid, kBlock, s, plan->rowptr, col, order,            \
             plan->long_rows, plan->chunk_ptr, el, er, x, g, out, rowmax, rowden, alpha, de, ger, pger,   \
             rng_used, d)
#define GGL_GAT_WIDE_H(V, LG)                                                                            \
  do {                                                                                                   \
    if (H == 1) GGL_GAT_WIDE(V, LG, 1);                                                                  \
    else if (H == 2) GGL_GAT_WIDE(V, LG, 2);                                                             \
    else if (H <= 4) GGL_GAT_WIDE(V, LG, 4);                                                             \
    else if (H <= 8) GGL_GAT_WIDE(V, LG, 8);                                                             \
    else GGL_GAT_WIDE(V, LG, 16);                                                                        \
  } while (0)
      if (vec4) {
        if (logg == 4) GGL_GAT_WIDE_H(4, 4);
        else if (logg == 5) GGL_GAT_WIDE_H(4, 5);
        else GGL_GAT_WIDE_H(4, 6);
      } else {
        if (logg <= 5) { logg = 5; GGL_GAT_WIDE_H(1, 5); }
        else GGL_GAT_WIDE_H(1, 6);
      }
#undef GGL_GAT_WIDE_H
#undef GGL_GAT_WIDE
      GGL_LAUNCH_CHECK();
      if (plan->n_long > 0) {
        GGL_LAUNCH((gat_bwd_dst_final_kernel), gat_grid_for(plan->n_long * H), kBlock, s, plan->long_rows,
                   plan->chunk_ptr, (const float *)pger, ger, plan->n_long, H);
        GGL_LAUNCH_CHECK();
      }
      return GGL_OK;
    }
  }
#endif
#define GGL_GAT_DST(V, CR)                                                                              \
  GGL_LAUNCH((gat_bwd_dst_kernel<V, CR>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,    \
             plan->chunk_ptr, el, er, x, g, out, rowmax, rowden, alpha, de, ger, pger, rng_used, d)
  if (vec4 && C == 8) GGL_GAT_DST(4, 8);
  else if (vec4 && C == 16) GGL_GAT_DST(4, 16);
  else if (vec4) GGL_GAT_DST(4, 0);
  else GGL_GAT_DST(1, 0);
#undef GGL_GAT_DST
  GGL_LAUNCH_CHECK();
  if (plan->n_long > 0) {
    GGL_LAUNCH((gat_bwd_dst_final_kernel), gat_grid_for(plan->n_long * H), kBlock, s, plan->long_rows,
               plan->chunk_ptr, (const float *)pger, ger, plan->n_long, H);
    GGL_LAUNCH_CHECK();
  }
  return GGL_OK;
}

// Source-major half of the backward (gat_bwd_src_kernel above).  planT->partial must hold
// ggl_partial_bytes(GGL_F32, n_chunks, H*C + H, 0) bytes when the transposed plan has long rows.
extern "C" int ggl_gat_fused_bwd_src(const ggl_segplan_t *planT, const int32_t *colT,
                                     const int32_t *posT, const float *alpha, const float *de,
                                     const float *g, int64_t H, int64_t C, float *gx, float *gel,
                                     void *stream) {
  GGL_REQUIRE(planT && planT->rowptr, GGL_EINVAL, "planT is NULL");
  GGL_REQUIRE(H > 0 && C > 0 && planT->chunk > 0, GGL_EINVAL, "H, C and chunk must be positive");
  const int64_t N = planT->N;
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(gx && gel, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((colT && posT && alpha && de && g) || planT->E == 0, GGL_EINVAL, "NULL pointer");
  GatDims d{};
  d.N = N; d.H = H; d.C = C; d.K = H * C; d.E = planT->E;
  d.es = (de == alpha + 1) ? 2 : 1;
  d.chunk = planT->chunk; d.n_long = planT->n_long; d.n_chunks = planT->n_chunks;
  float *pacc = nullptr, *pgel = nullptr;
  if (planT->n_long > 0) {
    GGL_REQUIRE(planT->long_rows && planT->chunk_ptr && planT->partial, GGL_EWORKSPACE,
                "transposed plan has long rows but long_rows/chunk_ptr/partial is NULL");
    pacc = static_cast<float *>(planT->partial);
    pgel = pacc + planT->n_chunks * d.K;
    d.chunk_blocks = ceil_div(planT->n_chunks, kWavesPerBlock);
  }
  const bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(gx) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(pacc) & 15u) == 0) && !options().force_generic;
  const int vec = vec4 ? 4 : 1;
  d.logL = pow2_log2(ceil_div(d.K, vec));
  d.nblocks = ceil_div(N, (int64_t)kWavesPerBlock * (kWave >> d.logL));
  const int64_t grid = d.chunk_blocks + d.nblocks;
  GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  const int32_t *order = options().row_order ? planT->row_order : nullptr;
  hipStream_t s = as_stream(stream);
  if (vec4)
    GGL_LAUNCH((gat_bwd_src_kernel<4>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
               planT->chunk_ptr, alpha, de, g, gx, gel, pacc, pgel, d);
  else
    GGL_LAUNCH((gat_bwd_src_kernel<1>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
               planT->chunk_ptr, alpha, de, g, gx, gel, pacc, pgel, d);
  GGL_LAUNCH_CHECK();
  if (planT->n_long > 0) {
    GGL_LAUNCH((gat_bwd_src_final_kernel), planT->n_long, kBlock, s, planT->long_rows, planT->chunk_ptr,
               (const float *)pacc, (const float *)pgel, gx, gel, d);
    GGL_LAUNCH_CHECK();
  }
  return GGL_OK;
}
