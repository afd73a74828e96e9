// gammagl_amd/csrc/gat_fast.hip — the GPU-only paths of the fused GAT op: kernels that exchange values between
// lanes (DPP moves, ds_bpermute) and therefore have no host-emulated build; the -m gpu suite is their checker
// (against the oracle through Engine.gat_fused, against the kernels of gat.hip, and at full size).  The
// host build links host/gpu_only_stubs.cpp in place of this file.
#include "gat_common.hpp"

namespace ggl {
// =====================================================================================================
// Fast path (GPU build only): heads of C = 4, 8, 16, 32 or 64 channels, H * C <= 256.
//
// The kernels above spend their time in the VECTOR ALU, not in memory: the ISA of gat_fwd_kernel<4,false>
// shows ~460 VALU instructions per 4-edge step (libm expf twice per edge, IEEE division, 64-bit address
// arithmetic per load, a predicated rescale per edge) against 12 memory instructions, and the Reddit-sized
// 60 MB feature panel is cache-resident — 114.8 M edges x 16 lanes of that is ~5 ms of issue time on its
// own.  These variants do the same walks with an order of magnitude fewer VALU instructions:
//   * exp through v_exp_f32 (exp2(x * log2 e), ~1e-6 relative on the operand range of a softmax, see fexp; the GAT
//     parity bar is 1e-5 relative against the oracle's three-pass restatement);
//   * one rescale per block of 4 / 8 edges (block maximum first) instead of a predicated one per edge;
//   * column indices as one 16-byte load per 4 edges (the walk is aligned to multiples of 4);
//   * 32-bit byte offsets into the feature panels where they fit (saddr + voffset addressing);
//   * FMA accumulation, reciprocal of the denominator once per row;
//   * work items (hub chunks first, then rows) all handled by a lane group of H * C / 4 lanes — hub chunks
//     used to get a whole wavefront of which K / 4 lanes worked.
// Backward without the [E, H, 2] alpha / de round trip (7.3 GB written by one walk and gathered through
// posT by the other on the Reddit-sized graph): both walks RECOMPUTE alpha and de from per-row constants —
//   destination walk (forward plan):  stats[i,h] = {er, m, 1/(den + 1e-16), <g_i, out_i>};  ger[i,h] = sum_p de
//   source walk (transposed plan):    gx[j,h,:] = sum_q alpha g_i,  gel[j,h] = sum_q de, with x_j, el_j in
//                                     registers and g_i (needed anyway) + stats[i,h] (a 30 MB panel) gathered
// with <g_i[h,:], x_j[h,:]> reduced over the C/4 lanes of a head by DPP / wave shuffles (gradients are held
// to 1e-5 relative, not bit-exact).  The host-emulated test build cannot shuffle between its sequentially
// executed lanes: it keeps the kernels above for every shape (they also remain the path for other shapes).
// =====================================================================================================
// exp(v), v <= 0 (a softmax exponent), through v_exp_f32: exp2(v * log2 e).  The scaling's rounding carries |v| * 6e-8 of
// relative error into the result (2e-7 at the typical v = -3, 1.7e-6 at v = -30).  Round 5 measured what that costs: with
// the scaling compensated (hi + lo product, five more instructions, 1.2e-7 everywhere) the layer's errors against an fp64
// evaluation moved from 1.44e-6 to 1.36e-6 (out) and 1.42e-6 to 1.23e-6 (gx), g_er not at all — and the Reddit step
// from 30.0 to 30.4 ms.  The exponential is NOT what limits the gradients' accuracy (the f32 row sums are: see
// gat_bwd_dst2_kernel); the plain form stays.
__device__ __forceinline__ float fexp(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896340736f); }

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the LPH (power of two, aligned) lanes of a head; every lane of the head gets the total
template <int LPH> __device__ __forceinline__ float head_sum(float v) {
  if (LPH >= 2) v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]: lane ^ 1
  if (LPH >= 4) v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]: lane ^ 2
  if (LPH >= 8) v += __shfl_xor(v, 4, 64);
  if (LPH >= 16) v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}

struct GatItem {
  int64_t row, beg, end, cid;
  bool is_chunk, first_chunk;
};
// work item -> (row, position range).  Items [0, n_chunks) are the chunks of long rows, then one per row slot.
__device__ __forceinline__ bool gat_item(const GatDims &d, const int64_t *__restrict__ rowptr,
                                         const int32_t *__restrict__ row_order,
                                         const int32_t *__restrict__ long_rows,
                                         const int64_t *__restrict__ chunk_ptr, int64_t item, GatItem &it) {
  it.is_chunk = item < d.n_chunks;
  it.cid = item;
  it.first_chunk = false;
  if (it.is_chunk) {
    int64_t lo = 0, hi = d.n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= item) lo = mid; else hi = mid - 1;
    }
    it.row = long_rows[lo];
    it.first_chunk = (item == chunk_ptr[lo]);
    it.beg = rowptr[it.row] + (item - chunk_ptr[lo]) * d.chunk;
    const int64_t rend = rowptr[it.row + 1];
    it.end = (it.beg + d.chunk < rend) ? it.beg + d.chunk : rend;
    return true;
  }
  const int64_t slot = item - d.n_chunks;
  it.row = row_order ? (int64_t)row_order[slot] : slot;
  it.beg = rowptr[it.row];
  it.end = rowptr[it.row + 1];
  return it.end - it.beg <= d.chunk;  // a long row is covered by its chunk items
}

// feature-panel addressing: byte offsets in 32 bits when the panel is < 4 GiB
template <bool OFF32> struct RowAddr {
  const char *base;
  uint32_t stride;  // bytes per row
  __device__ __forceinline__ const char *at(int32_t r) const {
    if (OFF32) return base + (uint32_t)((uint32_t)r * stride);
    return base + (int64_t)r * (int64_t)stride;
  }
};

#define GGL_GAT2_PROLOGUE()                                                                           \
  const int64_t item = thread_id() >> d.logL;                                                         \
  const int li = (int)threadIdx.x & ((1 << d.logL) - 1);                                              \
  if (item >= d.n_chunks + d.N) return;                                                               \
  GatItem it;                                                                                         \
  if (!gat_item(d, rowptr, row_order, long_rows, chunk_ptr, item, it)) return;                        \
  const int H = (int)d.H;                                                                             \
  const bool act = li < H * C4;      /* lanes past the last head idle along (H * C4 not a power of 2) */ \
  const int h = act ? li / C4 : 0;                                                                    \
  const int kk = act ? li * 4 : 0;                                                                    \
  const bool lead = act && (li % C4) == 0

template <int C4, bool DROP, bool OFF32>
__global__ __launch_bounds__(kBlock) void gat_fwd2_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ er,
    const float *__restrict__ x, float *__restrict__ y, float *__restrict__ rowmax,
    float *__restrict__ rowden, float *__restrict__ pacc, float *__restrict__ pm,
    float *__restrict__ pd, const int64_t *__restrict__ rng, const GatDims d) {
  GGL_GAT2_PROLOGUE();
  if (!act) return;  // no cross-lane traffic in the forward
  const int64_t K = d.K;
  const float slope = d.slope;
  const float er_i = er[it.row * H + h];
  const RowAddr<OFF32> xa{reinterpret_cast<const char *>(x) + kk * 4, (uint32_t)(K * 4)};
  const RowAddr<OFF32> ea{reinterpret_cast<const char *>(el) + h * 4, (uint32_t)(H * 4)};
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  float m = -FLT_MAX, den = 0.0f;
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  auto rescale = [&](float mx) {
    if (mx > m) {
      const float sc = fexp(m - mx);  // m = -FLT_MAX before the first edge: exp2(-inf) = 0
      den *= sc;
      a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
      m = mx;
    }
  };
  auto add = [&](float s, const float4 &v, uint32_t word) {
    const float w = fexp(s - m);
    den += w;
    float wk = w;
    if (DROP) wk = (word >= d.drop_thresh) ? w * d.drop_scale : 0.0f;
    a.x = __builtin_fmaf(v.x, wk, a.x); a.y = __builtin_fmaf(v.y, wk, a.y);
    a.z = __builtin_fmaf(v.z, wk, a.z); a.w = __builtin_fmaf(v.w, wk, a.w);
  };
  auto single = [&](int64_t q) {
    const int32_t c = col[q];
    const float4 v = *reinterpret_cast<const float4 *>(xa.at(c));
    const float s = lrelu(*reinterpret_cast<const float *>(ea.at(c)) + er_i, slope);
    rescale(s);
    add(s, v, DROP ? drop_word(q, H, h, offset, seed) : 0u);
  };
  int64_t p = it.beg;
  for (; p < it.end && (p & 3) != 0; ++p) single(p);
  for (; p + 8 <= it.end; p += 8) {  // 8 feature rows in flight, two 16-byte index loads
    const int4 i0 = *reinterpret_cast<const int4 *>(col + p), i1 = *reinterpret_cast<const int4 *>(col + p + 4);
    const int32_t c[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    float4 v[8];
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(xa.at(c[u]));
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = *reinterpret_cast<const float *>(ea.at(c[u]));
    float mx = -FLT_MAX;
#pragma unroll
    for (int u = 0; u < 8; ++u) { s[u] = lrelu(s[u] + er_i, slope); mx = fmaxf(mx, s[u]); }
    rescale(mx);
    U4 r0{0u, 0u, 0u, 0u}, r1{0u, 0u, 0u, 0u};
    if (DROP) {
      r0 = drop_words4(p >> 2, H, h, offset, seed);
      r1 = drop_words4((p >> 2) + 1, H, h, offset, seed);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add(s[u], v[u], pick_word(r0, u));
#pragma unroll
    for (int u = 4; u < 8; ++u) add(s[u], v[u], pick_word(r1, u - 4));
  }
  for (; p + 4 <= it.end; p += 4) {
    const int4 i0 = *reinterpret_cast<const int4 *>(col + p);
    const int32_t c[4] = {i0.x, i0.y, i0.z, i0.w};
    float4 v[4];
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(xa.at(c[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = *reinterpret_cast<const float *>(ea.at(c[u]));
    float mx = -FLT_MAX;
#pragma unroll
    for (int u = 0; u < 4; ++u) { s[u] = lrelu(s[u] + er_i, slope); mx = fmaxf(mx, s[u]); }
    rescale(mx);
    U4 r0{0u, 0u, 0u, 0u};
    if (DROP) r0 = drop_words4(p >> 2, H, h, offset, seed);
#pragma unroll
    for (int u = 0; u < 4; ++u) add(s[u], v[u], pick_word(r0, u));
  }
  for (; p < it.end; ++p) single(p);
  if (it.is_chunk) {
    *reinterpret_cast<float4 *>(pacc + it.cid * K + kk) = a;
    if (lead) {
      pm[it.cid * H + h] = m;
      pd[it.cid * H + h] = den;
    }
    return;
  }
  const float rinv = 1.0f / (den + 1e-16f);  // softmax.py:35: exp / (sum + 1e-16)
  a.x *= rinv; a.y *= rinv; a.z *= rinv; a.w *= rinv;
  *reinterpret_cast<float4 *>(y + it.row * K + kk) = a;
  if (lead) {
    rowmax[it.row * H + h] = m;
    rowden[it.row * H + h] = den;
  }
}

// destination walk of the backward: stats[i,h] = {er, m, 1 / (den + 1e-16), <g_i, out_i>} and
// ger[i,h] = sum_p de_p,  de_p = alpha_p (keep_p / (1 - p_drop) <g_i, x_j> - dot_i) LeakyReLU'(raw_p)
template <int C4, bool DROP, bool OFF32>
__global__ __launch_bounds__(kBlock) void gat_bwd_dst2_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ er,
    const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ out,
    const float *__restrict__ rowmax, const float *__restrict__ rowden, float *__restrict__ stats,
    float *__restrict__ ger, float *__restrict__ pger, const int64_t *__restrict__ rng, const GatDims d) {
  GGL_GAT2_PROLOGUE();
  const int64_t K = d.K;
  const float slope = d.slope;
  float4 gi = make_float4(0.0f, 0.0f, 0.0f, 0.0f), oi = gi;
  if (act) {
    gi = *reinterpret_cast<const float4 *>(g + it.row * K + kk);
    oi = *reinterpret_cast<const float4 *>(out + it.row * K + kk);
  }
  const float dot = head_sum<C4>(dot4(gi, oi));
  const float er_i = er[it.row * H + h];
  const float m = rowmax[it.row * H + h];
  const float rinv = 1.0f / (rowden[it.row * H + h] + 1e-16f);
  if (lead && (!it.is_chunk || it.first_chunk))
    *reinterpret_cast<float4 *>(stats + (it.row * H + h) * 4) = make_float4(er_i, m, rinv, dot);
  const RowAddr<OFF32> xa{reinterpret_cast<const char *>(x) + kk * 4, (uint32_t)(K * 4)};
  const RowAddr<OFF32> ea{reinterpret_cast<const char *>(el) + h * 4, (uint32_t)(H * 4)};
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  // g_er[i,h] = sum_p l'_p alpha_p (da_p - s_i), s_i = sum_q alpha_q da_q: a sum of thousands of cancelling terms whose
  // result is 10-100x smaller than they are.  With alpha = exp(.) / den in f32, sum_q alpha_q = 1 + eps (den is an f32 sum
  // of up to 10^5 exponentials) and the stored s_i = <g_i, out_i> carries the same eps: the row's result is off by
  // eps s_i sum_p l'_p alpha_p — 1.6e-5 of the tensor's magnitude against an fp64 evaluation of the layer where the
  // reference's own f32 composition has 5.5e-6 (tests/test_gpu_refsize.py).  Round 5: the row keeps FOUR sums in double,
  //   A = sum l' alpha da,  B = sum l' alpha,  S = sum alpha da,  W = sum alpha,   g_er = A - (S / W) B,
  // the weighted mean S / W normalised by the alphas actually used: eps cancels, what is left is the rounding of the f32
  // terms (numpy model of both forms over 10-20 000-edge rows: 10-20x smaller).  stats.w = <g_i, out_i> stays what the source
  // walk uses per edge (its errors are spread over different rows there).
  double sA = 0.0, sB = 0.0, sS = 0.0, sW = 0.0;
  auto edge = [&](float e, const float4 &v, uint32_t word) {
    float da = head_sum<C4>(dot4(gi, v));
    const float raw = e + er_i;
    const float al = fexp(lrelu(raw, slope) - m) * rinv;
    if (DROP) da = (word >= d.drop_thresh) ? da * d.drop_scale : 0.0f;  // d out / d alpha_p = keep / (1 - p) <g_i, x_j>
    const float lal = raw > 0.0f ? al : al * slope;
    sA += (double)(lal * da);
    sB += (double)lal;
    sS += (double)(al * da);
    sW += (double)al;
  };
  auto single = [&](int64_t q) {
    const int32_t c = col[q];
    const float4 v = *reinterpret_cast<const float4 *>(xa.at(c));
    edge(*reinterpret_cast<const float *>(ea.at(c)), v, DROP ? drop_word(q, H, h, offset, seed) : 0u);
  };
  int64_t p = it.beg;
  for (; p < it.end && (p & 3) != 0; ++p) single(p);
  for (; p + 8 <= it.end; p += 8) {
    const int4 i0 = *reinterpret_cast<const int4 *>(col + p), i1 = *reinterpret_cast<const int4 *>(col + p + 4);
    const int32_t c[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    float4 v[8];
    float e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(xa.at(c[u]));
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = *reinterpret_cast<const float *>(ea.at(c[u]));
    U4 r0{0u, 0u, 0u, 0u}, r1{0u, 0u, 0u, 0u};
    if (DROP) {
      r0 = drop_words4(p >> 2, H, h, offset, seed);
      r1 = drop_words4((p >> 2) + 1, H, h, offset, seed);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) edge(e[u], v[u], pick_word(r0, u));
#pragma unroll
    for (int u = 4; u < 8; ++u) edge(e[u], v[u], pick_word(r1, u - 4));
  }
  for (; p + 4 <= it.end; p += 4) {
    const int4 i0 = *reinterpret_cast<const int4 *>(col + p);
    const int32_t c[4] = {i0.x, i0.y, i0.z, i0.w};
    float4 v[4];
    float e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(xa.at(c[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = *reinterpret_cast<const float *>(ea.at(c[u]));
    U4 r0{0u, 0u, 0u, 0u};
    if (DROP) r0 = drop_words4(p >> 2, H, h, offset, seed);
#pragma unroll
    for (int u = 0; u < 4; ++u) edge(e[u], v[u], pick_word(r0, u));
  }
  for (; p < it.end; ++p) single(p);
  if (lead) {
    if (it.is_chunk) {      // a hub row's chunk: the four sums, combined in chunk order by gat_bwd_dst_final4_kernel
      double *pq = reinterpret_cast<double *>(pger) + (it.cid * H + h) * 4;
      pq[0] = sA; pq[1] = sB; pq[2] = sS; pq[3] = sW;
    } else {
      ger[it.row * H + h] = sW > 0.0 ? (float)(sA - (sS / sW) * sB) : 0.0f;
    }
  }
}

// long rows of the destination walk above: the four double sums of a row's chunks, in chunk order
__global__ __launch_bounds__(kBlock) void gat_bwd_dst_final4_kernel(const int32_t *__restrict__ long_rows,
                                                                    const int64_t *__restrict__ chunk_ptr,
                                                                    const double *__restrict__ pq, float *__restrict__ ger,
                                                                    int64_t n_long, int64_t H) {
  const int64_t stride = grid_threads();
  for (int64_t t = thread_id(); t < n_long * H; t += stride) {
    const int64_t j = t / H, h = t - j * H;
    double sA = 0.0, sB = 0.0, sS = 0.0, sW = 0.0;
    for (int64_t c = chunk_ptr[j]; c < chunk_ptr[j + 1]; ++c) {
      const double *q = pq + (c * H + h) * 4;
      sA += q[0]; sB += q[1]; sS += q[2]; sW += q[3];
    }
    ger[(int64_t)long_rows[j] * H + h] = sW > 0.0 ? (float)(sA - (sS / sW) * sB) : 0.0f;
  }
}

// source walk of the backward (transposed plan; row = source node j, d.N = number of source rows):
//   gx[j,h,:] = sum_q alpha_q keep_q / (1 - p_drop) g[i_q,h,:],   gel[j,h] = sum_q de_q,   i_q = colT[q]
template <int C4, bool DROP, bool OFF32>
__global__ __launch_bounds__(kBlock) void gat_bwd_src2_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col /* colT */,
    const int32_t *__restrict__ posT, const int32_t *__restrict__ row_order,
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr,
    const float *__restrict__ el, const float *__restrict__ x, const float *__restrict__ g,
    const float *__restrict__ stats, float *__restrict__ gx, float *__restrict__ gel,
    float *__restrict__ pacc, float *__restrict__ pgel, const int64_t *__restrict__ rng, const GatDims d) {
  GGL_GAT2_PROLOGUE();
  const int64_t K = d.K;
  const float slope = d.slope;
  float4 xj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (act) xj = *reinterpret_cast<const float4 *>(x + it.row * K + kk);
  const float el_j = el[it.row * H + h];
  const RowAddr<OFF32> ga{reinterpret_cast<const char *>(g) + kk * 4, (uint32_t)(K * 4)};
  const RowAddr<OFF32> sa{reinterpret_cast<const char *>(stats) + h * 16, (uint32_t)(H * 16)};
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float gl = 0.0f;
  auto edge = [&](const float4 &gi, const float4 &st, int32_t fq /* forward position of the edge */) {
    float da = head_sum<C4>(dot4(gi, xj));
    const float raw = el_j + st.x;
    const float al = fexp(lrelu(raw, slope) - st.y) * st.z;
    float alk = al;
    if (DROP) {  // the keep bit of the FORWARD position of this edge
      const bool keep = drop_word((int64_t)fq, H, h, offset, seed) >= d.drop_thresh;
      alk = keep ? al * d.drop_scale : 0.0f;
      da = keep ? da * d.drop_scale : 0.0f;
    }
    const float ds = al * (da - st.w);
    gl += raw > 0.0f ? ds : ds * slope;
    a.x = __builtin_fmaf(gi.x, alk, a.x); a.y = __builtin_fmaf(gi.y, alk, a.y);
    a.z = __builtin_fmaf(gi.z, alk, a.z); a.w = __builtin_fmaf(gi.w, alk, a.w);
  };
  auto single = [&](int64_t q) {
    const int32_t c = col[q];
    const float4 gi = *reinterpret_cast<const float4 *>(ga.at(c));
    const float4 st = *reinterpret_cast<const float4 *>(sa.at(c));
    edge(gi, st, DROP ? posT[q] : 0);
  };
  int64_t p = it.beg;
  for (; p < it.end && (p & 3) != 0; ++p) single(p);
  for (; p + 4 <= it.end; p += 4) {  // 4 gradient rows + 4 stats vectors in flight
    const int4 i0 = *reinterpret_cast<const int4 *>(col + p);
    const int32_t c[4] = {i0.x, i0.y, i0.z, i0.w};
    int4 f0 = make_int4(0, 0, 0, 0);  // the block's forward positions (dropout), fetched beside the ids
    if (DROP) f0 = *reinterpret_cast<const int4 *>(posT + p);
    const int32_t fq[4] = {f0.x, f0.y, f0.z, f0.w};
    float4 gi[4], st[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) gi[u] = *reinterpret_cast<const float4 *>(ga.at(c[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) st[u] = *reinterpret_cast<const float4 *>(sa.at(c[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) edge(gi[u], st[u], fq[u]);
  }
  for (; p < it.end; ++p) single(p);
  if (!act) return;
  if (it.is_chunk) {
    *reinterpret_cast<float4 *>(pacc + it.cid * K + kk) = a;
    if (lead) pgel[it.cid * H + h] = gl;
    return;
  }
  *reinterpret_cast<float4 *>(gx + it.row * K + kk) = a;
  if (lead) gel[it.row * H + h] = gl;
}


// =====================================================================================================
// Head-mean GAT layer with a SHARED input row (GPU build only): the output layer of a GAT
// (models/gat.py: concat=False, gat_conv.py:114-122 averages its heads) aggregated BEFORE it is transformed.
//
// The layer computes y_i = 1/H sum_h sum_j alpha_ijh (x_j W_h).  Aggregation is linear, so
//     y_i = 1/H sum_h (sum_j alpha_ijh x_j) W_h = 1/H A_i[H*F] @ Wst[H*F, C],   A_ih = sum_j alpha_ijh x_j,
// and the logits need no transformed rows either: el = x @ U, U[f,h] = sum_c W[f,h,c] a_src[h,c].  For the Reddit
// GAT (F = 64 hidden, H = 8, C = 41) the per-edge gather shrinks from the 8 x 44-float transformed row (1408 B,
// three walks at the roofline of those rows: 20 + 22 + 21 ms) to the 64-float input row (256 B); the backward
// exploits the head mean the same way — dL/dA_ih = (g_i / H) W_h^T has rank-1 structure per row, so the source
// walk gathers g_i (C floats) and takes its dots against the row's OWN transformed features:
//     <dA_ih, x_j> = <g_i / H, x_j W_h>.
// Every gather of the three walks is <= 256 B + 128 B of row constants per edge.
//
// Lane layout (all kernels): 16 lanes (one DPP row) per work item (row or hub chunk), H = 8 heads.
//   * as a CHANNEL lane, lane l owns floats [4l, 4l+4) of the gathered row and of all 8 per-head accumulators;
//   * as a WEIGHT lane, lane l = 8e + h computes the scalar softmax terms of head h for the e-th edge of a pair;
//     weights reach the channel lanes by `row_newbcast` (one DPP move per head and edge), per-head dot products
//     reach the weight lanes by a 16-value reduce-scatter over the row (row_ror:8, xor 4, quad_perm xor 2 / 1).
//   * the row maximum of the logits is found by a first pass over (col, el) alone — 36 B per edge from a
//     cache-resident panel — so the main walk needs no online rescaling.
// =====================================================================================================
constexpr int kShH = 8;  // heads (fixed: 16 lanes = 2 edges x 8 heads)

template <int N> __device__ __forceinline__ float row_bcast(float v) {  // value of lane N of this 16-lane row
  return dpp_mov<0x150 + N>(v);
}
__device__ __forceinline__ float row_ror8(float v) { return dpp_mov<0x128>(v); }  // lane ^ 8 of the row

__device__ __forceinline__ double row_ror8_f64(double v) {   // lane ^ 8 of the row, both halves of the double
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), 0x128, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x128, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}

// v[k] (k = 0..15) summed over the 16 lanes of the row; lane l returns the total of v[l]
__device__ __forceinline__ float reduce_scatter16(const float (&v)[16], int li) {
  const bool b3 = li & 8, b2 = li & 4, b1 = li & 2, b0 = li & 1;
  float a[8], b[4], c[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (b3 ? v[8 + k] : v[k]) + row_ror8(b3 ? v[k] : v[8 + k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = (b2 ? a[4 + k] : a[k]) + __shfl_xor(b2 ? a[k] : a[4 + k], 4, 64);
#pragma unroll
  for (int k = 0; k < 2; ++k) c[k] = (b1 ? b[2 + k] : b[k]) + dpp_mov<0x4E>(b1 ? b[k] : b[2 + k]);
  return (b0 ? c[1] : c[0]) + dpp_mov<0xB1>(b0 ? c[0] : c[1]);
}

// broadcast the 8 per-head weights of edge slot E (weight lanes 8E .. 8E+7) and accumulate w_h * x into acc[h]
template <int E> __device__ __forceinline__ void sh_accumulate(float wk, const float4 &x, float4 (&acc)[kShH]) {
#define GGL_SH_ACC(HH)                                                                 \
  {                                                                                    \
    const float w = row_bcast<8 * E + HH>(wk);                                         \
    acc[HH].x = __builtin_fmaf(x.x, w, acc[HH].x); acc[HH].y = __builtin_fmaf(x.y, w, acc[HH].y); \
    acc[HH].z = __builtin_fmaf(x.z, w, acc[HH].z); acc[HH].w = __builtin_fmaf(x.w, w, acc[HH].w); \
  }
  GGL_SH_ACC(0) GGL_SH_ACC(1) GGL_SH_ACC(2) GGL_SH_ACC(3) GGL_SH_ACC(4) GGL_SH_ACC(5) GGL_SH_ACC(6) GGL_SH_ACC(7)
#undef GGL_SH_ACC
}

// ---- round 6: the 16 dots of an edge pair as PACKED products over head pairs, reduced without per-step selects ----------
// The walks of the backward take, per pair of edges (A, B) and per lane, 8 heads x 2 edges dot products of the lane's 4
// columns, then reduce-scatter the 16 partials over the 16 lanes (lane 8e + h ends with the total for edge e, head h).  ISA of
// round 5's form, per 4-edge step of the source walk: 444 VALU instructions, of which 128 scalar multiply / FMAs for the dots
// and 60 v_cndmask + 30 DPP / swizzle adds for the two reduce-scatters (each stage selected "my half" / "the other half" of
// its values per lane) — the kernel is VALU-bound (26 wave-instructions per edge x 114.8 M edges / 614 G wave-instructions per
// second = 4.9 ms of its 7.5), not latency-bound as round 5 read it.  Now:
//   * the row's constant operand (z_j of the source walk, G_i of the destination walk) is kept per lane as head PAIRS
//     P[c][qp] = (R[2 qp].c, R[2 qp + 1].c), so a dot step is one v_pk_fma_f32 for two heads (the gathered value enters through
//     op_sel as a splat): 64 packed instead of 128 scalar FMAs per step;
//   * each lane holds the rows of its constant operand PERMUTED by its own lane id, R[q] = Z[q ^ (lane & 7)] (a permuted load
//     address, once per row): the three head-splitting stages of the reduce-scatter then need no selects at all — "my" value is
//     always slot k, "the other lane's" always slot k + half.  Only the edge-splitting stage selects, on the two gathered float4s
//     (8 v_cndmask per pair instead of 30).
// Same sums in another association: results move within f32 rounding (tests: fp64 truth, test_gpu_refsize.py).
typedef float f2v __attribute__((ext_vector_type(2)));
struct ShPairs { f2v p[4][4]; };   // [column of the lane's float4][head pair]

// P of this lane from an [8, F] row panel (`panel` = row base + the lane's column offset)
__device__ __forceinline__ void sh_pairs_load(ShPairs &P, const float *__restrict__ panel, int64_t F, int li, bool act) {
  float4 t[kShH];
#pragma unroll
  for (int q = 0; q < kShH; ++q)
    t[q] = act ? *reinterpret_cast<const float4 *>(panel + (int64_t)(q ^ (li & 7)) * F) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int qp = 0; qp < 4; ++qp) {
    P.p[0][qp] = f2v{t[2 * qp].x, t[2 * qp + 1].x};
    P.p[1][qp] = f2v{t[2 * qp].y, t[2 * qp + 1].y};
    P.p[2][qp] = f2v{t[2 * qp].z, t[2 * qp + 1].z};
    P.p[3][qp] = f2v{t[2 * qp].w, t[2 * qp + 1].w};
  }
}
// per-lane LDS slots of P (8 float4 per lane: slot 2 c + j holds head pairs 2 j, 2 j + 1 of column c)
template <typename L> __device__ __forceinline__ void sh_pairs_to_lds(const ShPairs &P, L &lds) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      lds[2 * c + j][threadIdx.x] = make_float4(P.p[c][2 * j].x, P.p[c][2 * j].y, P.p[c][2 * j + 1].x, P.p[c][2 * j + 1].y);
}
// the pair's 16 partial dots of this lane: v0[qp] = (head 2 qp, 2 qp + 1) against g0, v1 against g1
template <bool LDS, typename L>
__device__ __forceinline__ void sh_pair_dots(const ShPairs &P, L &lds, const float4 &g0, const float4 &g1, f2v (&v0)[4],
                                             f2v (&v1)[4]) {
  const float ga[4] = {g0.x, g0.y, g0.z, g0.w}, gb[4] = {g1.x, g1.y, g1.z, g1.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f2v pc[4];
    if constexpr (LDS) {
      const float4 lo = lds[2 * c][threadIdx.x], hi = lds[2 * c + 1][threadIdx.x];
      pc[0] = f2v{lo.x, lo.y}; pc[1] = f2v{lo.z, lo.w}; pc[2] = f2v{hi.x, hi.y}; pc[3] = f2v{hi.z, hi.w};
    } else {
#pragma unroll
      for (int qp = 0; qp < 4; ++qp) pc[qp] = P.p[c][qp];
    }
    const f2v sa = f2v{ga[c], ga[c]}, sb = f2v{gb[c], gb[c]};
#pragma unroll
    for (int qp = 0; qp < 4; ++qp) {
      v0[qp] = c == 0 ? pc[qp] * sa : __builtin_elementwise_fma(pc[qp], sa, v0[qp]);
      v1[qp] = c == 0 ? pc[qp] * sb : __builtin_elementwise_fma(pc[qp], sb, v1[qp]);
    }
  }
}
// reduce-scatter of the 16 partials over the row's 16 lanes; lane 8 e + h returns the total of (its slot-0 edge, head h).
// Slot k of lane l holds head k ^ (l & 7) (sh_pairs_load's permutation) and lanes >= 8 hold the pair's edges swapped.
__device__ __forceinline__ float sh_pair_reduce(const f2v (&v0)[4], const f2v (&v1)[4]) {
  float a[8], b[4], c[2];
#pragma unroll
  for (int qp = 0; qp < 4; ++qp) {
    a[2 * qp] = v0[qp].x + row_ror8(v1[qp].x);
    a[2 * qp + 1] = v0[qp].y + row_ror8(v1[qp].y);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = a[k] + __shfl_xor(a[4 + k], 4, 64);
#pragma unroll
  for (int k = 0; k < 2; ++k) c[k] = b[k] + dpp_mov<0x4E>(b[2 + k]);
  return c[0] + dpp_mov<0xB1>(c[1]);
}
__device__ __forceinline__ float4 sel4(bool s, const float4 &a, const float4 &b) {
  return make_float4(s ? a.x : b.x, s ? a.y : b.y, s ? a.z : b.z, s ? a.w : b.w);
}

struct ShDims {
  float slope;
  int64_t N, F, E;        // rows of this walk, floats per gathered row (<= 64, multiple of 4)
  int64_t chunk, n_long, n_chunks;
  uint32_t drop_thresh;
  float drop_scale;
};

#define GGL_SH_PROLOGUE()                                                                        \
  const int64_t item = thread_id() >> 4;                                                         \
  const int li = (int)threadIdx.x & 15;                                                          \
  if (item >= d.n_chunks + d.N) return;                                                          \
  GatDims gd{};                                                                                  \
  gd.N = d.N; gd.chunk = d.chunk; gd.n_long = d.n_long; gd.n_chunks = d.n_chunks;                \
  GatItem it;                                                                                    \
  if (!gat_item(gd, rowptr, row_order, long_rows, chunk_ptr, item, it)) return;                  \
  const int e = li >> 3, h = li & 7;                                                             \
  const bool act = 4 * li < (int)d.F; /* channel lanes past the row width idle along: they gather columns 0-3 again (kk = 0) \
     instead of being zero-filled — 16 v_mov + an exec-mask dance per 4-edge step (round 6) — which is harmless because the row's \
     constant operand IS zero-filled for them (their dots vanish) and nothing they accumulate is stored */ \
  const int kk = act ? 4 * li : 0

// pass 0: m[i,h] = max_p LeakyReLU(el[col[p],h] + er[i,h])   (-FLT_MAX for an empty row: unsorted_segment_max)
__global__ __launch_bounds__(kBlock) void gat_sh_rowmax_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ row_order,
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el,
    const float *__restrict__ er, float *__restrict__ rowmax, float *__restrict__ pmax, const ShDims d) {
  GGL_SH_PROLOGUE();
  (void)kk;
  const float er_i = er[it.row * kShH + h];
  float m = -FLT_MAX;
  int64_t p = it.beg + e;
  for (; p + 6 < it.end; p += 8) {  // 4 of this lane's edges in flight
    const int32_t c0 = col[p], c1 = col[p + 2], c2 = col[p + 4], c3 = col[p + 6];
    const float s0 = el[(int64_t)c0 * kShH + h], s1 = el[(int64_t)c1 * kShH + h], s2 = el[(int64_t)c2 * kShH + h],
                s3 = el[(int64_t)c3 * kShH + h];
    m = fmaxf(fmaxf(m, lrelu(s0 + er_i, d.slope)), lrelu(s1 + er_i, d.slope));
    m = fmaxf(fmaxf(m, lrelu(s2 + er_i, d.slope)), lrelu(s3 + er_i, d.slope));
  }
  for (; p < it.end; p += 2) m = fmaxf(m, lrelu(el[(int64_t)col[p] * kShH + h] + er_i, d.slope));
  m = fmaxf(m, row_ror8(m));
  if (e == 0) {
    if (it.is_chunk) pmax[it.cid * kShH + h] = m;
    else rowmax[it.row * kShH + h] = m;
  }
}
__global__ __launch_bounds__(kBlock) void gat_sh_rowmax_final_kernel(const int32_t *__restrict__ long_rows,
                                                                     const int64_t *__restrict__ chunk_ptr,
                                                                     const float *__restrict__ pmax,
                                                                     float *__restrict__ rowmax, int64_t n_long) {
  const int64_t stride = grid_threads();
  for (int64_t t = thread_id(); t < n_long * kShH; t += stride) {
    const int64_t j = t / kShH, h = t - j * kShH;
    float m = -FLT_MAX;
    for (int64_t c = chunk_ptr[j]; c < chunk_ptr[j + 1]; ++c) m = fmaxf(m, pmax[c * kShH + h]);
    rowmax[(int64_t)long_rows[j] * kShH + h] = m;
  }
}

// gather of 4 consecutive positions with validity (blocks at the ends of a row are partly outside it)
struct ShBlock {
  int32_t c[4];
  bool ok[4];
};
__device__ __forceinline__ void sh_block(const int32_t *__restrict__ col, int64_t p0, int64_t lo, int64_t hi, ShBlock &b) {
  if (p0 >= lo && p0 + 4 <= hi) {
    const int4 t = *reinterpret_cast<const int4 *>(col + p0);
    b.c[0] = t.x; b.c[1] = t.y; b.c[2] = t.z; b.c[3] = t.w;
    b.ok[0] = b.ok[1] = b.ok[2] = b.ok[3] = true;
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      b.ok[u] = p0 + u >= lo && p0 + u < hi;
      b.c[u] = b.ok[u] ? col[p0 + u] : 0;
    }
  }
}

// forward: A[i,h,:] = sum_p keep_p/(1-pd) exp(s_p - m) x[col[p],:] / (den + 1e-16), den[i,h] = sum_p exp(s_p - m)
// PF (round 5, with gat_sh_bwd_src's ZLDS form: option gat_sh_zlds): the NEXT step's column ids are requested before this step's
// gathers — a step then waits for one memory round trip instead of two dependent ones
template <bool DROP, bool PF = false>
__global__ __launch_bounds__(kBlock) void gat_sh_fwd_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ row_order,
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el,
    const float *__restrict__ er, const float *__restrict__ rowmax, const float *__restrict__ x,
    float *__restrict__ A, float *__restrict__ den_out, float *__restrict__ pacc, float *__restrict__ pden,
    const int64_t *__restrict__ rng, const ShDims d) {
  GGL_SH_PROLOGUE();
  const int64_t F = d.F;
  const float er_i = er[it.row * kShH + h], m = rowmax[it.row * kShH + h];
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  float4 acc[kShH];
#pragma unroll
  for (int q = 0; q < kShH; ++q) acc[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float den = 0.0f;
  ShBlock b, nb;
  if (PF && (it.beg & ~(int64_t)3) < it.end) sh_block(col, it.beg & ~(int64_t)3, it.beg, it.end, nb);
  for (int64_t p0 = it.beg & ~(int64_t)3; p0 < it.end; p0 += 4) {
    if (PF) {
      b = nb;
      if (p0 + 4 < it.end) sh_block(col, p0 + 4, it.beg, it.end, nb);
    } else {
      sh_block(col, p0, it.beg, it.end, b);
    }
    float4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + (int64_t)b.c[u] * F + kk);   // (idle channel lanes re-read columns 0-3: kk = 0; their sums are never stored)
    // this weight lane's two edges of the block: u = e and u = e + 2
    const int32_t cA = e ? b.c[1] : b.c[0], cB = e ? b.c[3] : b.c[2];
    const bool okA = e ? b.ok[1] : b.ok[0], okB = e ? b.ok[3] : b.ok[2];
    const float s0 = el[(int64_t)cA * kShH + h], s1 = el[(int64_t)cB * kShH + h];
    float w0 = okA ? fexp(lrelu(s0 + er_i, d.slope) - m) : 0.0f;
    float w1 = okB ? fexp(lrelu(s1 + er_i, d.slope) - m) : 0.0f;
    den += w0 + w1;
    if (DROP) {
      const U4 rw = drop_words4(p0 >> 2, kShH, h, offset, seed);
      w0 = (pick_word(rw, e) >= d.drop_thresh) ? w0 * d.drop_scale : 0.0f;
      w1 = (pick_word(rw, e + 2) >= d.drop_thresh) ? w1 * d.drop_scale : 0.0f;
    }
    sh_accumulate<0>(w0, xv[0], acc);
    sh_accumulate<1>(w0, xv[1], acc);
    sh_accumulate<0>(w1, xv[2], acc);
    sh_accumulate<1>(w1, xv[3], acc);
  }
  den += row_ror8(den);  // both edge parities of head h
  if (it.is_chunk) {
    if (act) {
#pragma unroll
      for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(pacc + (it.cid * kShH + q) * F + kk) = acc[q];
    }
    if (e == 0) pden[it.cid * kShH + h] = den;
    return;
  }
  const float rinv = 1.0f / (den + 1e-16f);
#define GGL_SH_NORM(HH)                                                                          \
  {                                                                                              \
    const float r = row_bcast<HH>(rinv);                                                         \
    acc[HH].x *= r; acc[HH].y *= r; acc[HH].z *= r; acc[HH].w *= r;                              \
  }
  GGL_SH_NORM(0) GGL_SH_NORM(1) GGL_SH_NORM(2) GGL_SH_NORM(3) GGL_SH_NORM(4) GGL_SH_NORM(5) GGL_SH_NORM(6) GGL_SH_NORM(7)
#undef GGL_SH_NORM
  if (act) {
#pragma unroll
    for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(A + (it.row * kShH + q) * F + kk) = acc[q];
  }
  if (e == 0) den_out[it.row * kShH + h] = den;
}

// long rows of the forward: A = sum_c pacc_c / (sum_c pden_c + 1e-16) (the maximum is global: partials just add)
__global__ __launch_bounds__(kBlock) void gat_sh_fwd_final_kernel(const int32_t *__restrict__ long_rows,
                                                                  const int64_t *__restrict__ chunk_ptr,
                                                                  const float *__restrict__ pacc,
                                                                  const float *__restrict__ pden,
                                                                  float *__restrict__ A, float *__restrict__ den_out,
                                                                  int64_t n_long, int64_t F) {
  const int64_t j = block_id();
  if (j >= n_long) return;
  const int64_t row = long_rows[j], c0 = chunk_ptr[j], c1 = chunk_ptr[j + 1], K = kShH * F;
  for (int64_t k = threadIdx.x; k < K; k += kBlock) {
    const int64_t hh = k / F;
    float a = 0.0f, dn = 0.0f;
    for (int64_t c = c0; c < c1; ++c) {
      a += pacc[c * K + k];
      dn += pden[c * kShH + hh];
    }
    A[row * K + k] = a / (dn + 1e-16f);
    if (k == hh * F) den_out[row * kShH + hh] = dn;
  }
}

// destination walk of the backward: ger[i,h] = sum_p de_p with <G_i[h,:], x_j> from the row's G in registers
// GLDS (option gat_sh_glds, A/B): the row's G (8 heads x this lane's 4 columns, dot-product operands only) in per-lane LDS slots
// like gat_sh_bwd_src's z_j
// PK (round 6, option gat_sh_pk): the dots packed over head pairs and reduced without selects (sh_pair_dots / sh_pair_reduce)
template <bool DROP, bool PF = false, bool GLDS = false, bool PK = false>
__global__ __launch_bounds__(kBlock) void gat_sh_bwd_dst_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ row_order,
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el,
    const float *__restrict__ x, const float *__restrict__ G, const float *__restrict__ stats,
    float *__restrict__ ger, float *__restrict__ pger, const int64_t *__restrict__ rng, const ShDims d) {
  GGL_SH_PROLOGUE();
  const int64_t F = d.F;
  __shared__ float4 gs_lds[GLDS ? kShH : 1][GLDS ? kBlock : 1];
  float4 g[(GLDS || PK) ? 1 : kShH];
  ShPairs GP;
  if (PK) {
    sh_pairs_load(GP, G + it.row * kShH * F + kk, F, li, act);
    if constexpr (GLDS) sh_pairs_to_lds(GP, gs_lds);
  } else {
#pragma unroll
    for (int q = 0; q < kShH; ++q) {
      const float4 gq = act ? *reinterpret_cast<const float4 *>(G + (it.row * kShH + q) * F + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (GLDS) gs_lds[q][threadIdx.x] = gq; else g[(GLDS || PK) ? 0 : q] = gq;
    }
  }
  const float4 st = *reinterpret_cast<const float4 *>(stats + (it.row * kShH + h) * 4);  // {er, m, rinv, dot}
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  // g_er[i,h] = sum_p l'_p alpha_p (da_p - s_i) is a sum of up to 10^5 cancelling terms; with the stored s_i = <G_i, A_i> and f32
  // sums the row's result carried the eps of `sum alpha = 1 + eps` (gat_bwd_dst2_kernel's comment: the hidden-layer kernels keep
  // FOUR sums in double since round 5).  Round 6: the same four sums here — A = sum l' alpha da, B = sum l' alpha, S = sum alpha da,
  // W = sum alpha, g_er = A - (S / W) B — after the full-size comparison with the transform-first kernels showed this walk's f32
  // sum 1.5e-2 of a row's scale apart on 10^5-edge rows (profiles/r6_gat_fullsize_forms.txt).  stats.w is no longer read here.
  double sA = 0.0, sB = 0.0, sS = 0.0, sW = 0.0;
  ShBlock b, nb;
  if (PF && (it.beg & ~(int64_t)3) < it.end) sh_block(col, it.beg & ~(int64_t)3, it.beg, it.end, nb);
  for (int64_t p0 = it.beg & ~(int64_t)3; p0 < it.end; p0 += 4) {
    if (PF) {
      b = nb;
      if (p0 + 4 < it.end) sh_block(col, p0 + 4, it.beg, it.end, nb);
    } else {
      sh_block(col, p0, it.beg, it.end, b);
    }
    float4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + (int64_t)b.c[u] * F + kk);   // (idle channel lanes re-read columns 0-3: kk = 0; their sums are never stored)
    const int32_t cA = e ? b.c[1] : b.c[0], cB = e ? b.c[3] : b.c[2];
    const bool okA = e ? b.ok[1] : b.ok[0], okB = e ? b.ok[3] : b.ok[2];
    const float s0 = el[(int64_t)cA * kShH + h], s1 = el[(int64_t)cB * kShH + h];
    U4 rw{0u, 0u, 0u, 0u};
    if (DROP) rw = drop_words4(p0 >> 2, kShH, h, offset, seed);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {  // pair of edges (2 pr, 2 pr + 1): 16 dots -> one per weight lane
      float da;
      if (GLDS) asm volatile("" ::: "memory");   // (re-read every step: hoisted, the slots would be registers again)
      if (PK) {
        f2v v0[4], v1[4];
        sh_pair_dots<GLDS>(GP, gs_lds, sel4(e != 0, xv[2 * pr + 1], xv[2 * pr]), sel4(e != 0, xv[2 * pr], xv[2 * pr + 1]), v0, v1);
        da = sh_pair_reduce(v0, v1);
      } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < kShH; ++q) {
          const float4 gq = GLDS ? gs_lds[q][threadIdx.x] : g[(GLDS || PK) ? 0 : q];
          v[q] = dot4(gq, xv[2 * pr]);
          v[8 + q] = dot4(gq, xv[2 * pr + 1]);
        }
        da = reduce_scatter16(v, li);
      }
      const float raw = (pr ? s1 : s0) + st.x;
      const float al = fexp(lrelu(raw, d.slope) - st.y) * st.z;
      if (DROP) da = (pick_word(rw, e + 2 * pr) >= d.drop_thresh) ? da * d.drop_scale : 0.0f;
      const float alo = (pr ? okB : okA) ? al : 0.0f;
      const float lal = raw > 0.0f ? alo : alo * d.slope;
      sA += (double)(lal * da);
      sB += (double)lal;
      sS += (double)(alo * da);
      sW += (double)alo;
    }
  }
  sA += row_ror8_f64(sA); sB += row_ror8_f64(sB); sS += row_ror8_f64(sS); sW += row_ror8_f64(sW);   // both edge parities of head h
  if (e == 0) {
    if (it.is_chunk) {   // a hub row's chunk: the four sums, combined in chunk order by gat_bwd_dst_final4_kernel
      double *pq = reinterpret_cast<double *>(pger) + (it.cid * kShH + h) * 4;
      pq[0] = sA; pq[1] = sB; pq[2] = sS; pq[3] = sW;
    } else {
      ger[it.row * kShH + h] = sW > 0.0 ? (float)(sA - (sS / sW) * sB) : 0.0f;
    }
  }
}

// source walk of the backward (transposed plan): T[j,h,:] = sum_q alpha_q keep_q/(1-pd) gy[i_q,:] and
// gel[j,h] = sum_q de_q, with <dA_ih, x_j> = <gy_i, z_jh>, z_j = the row's own transformed features (registers)
// WAVES: wavefronts per SIMD the register allocator is asked to leave room for (1 = no request).  With dropout the kernel
// needs 140 registers = 3 wavefronts per SIMD (without: 127 = 4); asked for 4 it fits 128 with 10 spilled values (44 bytes of
// scratch).  Option `gat_sh_waves` (A/B, round 5): profiles/r5_gat_sh_waves.txt.
// ZLDS (round 5, A/B: option gat_sh_zlds): the row's own transformed features z_j (8 heads x this lane's 4 columns = 32
// registers that only feed the dot products) live in LDS instead — each lane reads back exactly the slot it wrote, so no barrier
// is involved: LDS as a per-lane register file.  140 -> ~110 registers = 4 wavefronts per SIMD without spills; costs 16
// ds_read_b128 per 4-edge step (the LDS pipe is otherwise idle here).
template <bool DROP, int WAVES = 1, bool ZLDS = false, bool PK = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void gat_sh_bwd_src_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col /* colT */, const int32_t *__restrict__ posT,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ z,
    const float *__restrict__ gy, const float *__restrict__ stats, float *__restrict__ T, float *__restrict__ gel,
    float *__restrict__ pacc, float *__restrict__ pgel, const int64_t *__restrict__ rng, const ShDims d) {
  GGL_SH_PROLOGUE();
  const int64_t F = d.F;  // here: padded class width of gy / z rows
  __shared__ float4 zs[ZLDS ? kShH : 1][ZLDS ? kBlock : 1];
  float4 zr[(ZLDS || PK) ? 1 : kShH], acc[kShH];
  ShPairs ZP;
  if (PK) {
    sh_pairs_load(ZP, z + it.row * kShH * F + kk, F, li, act);
    if constexpr (ZLDS) sh_pairs_to_lds(ZP, zs);
  }
#pragma unroll
  for (int q = 0; q < kShH; ++q) {
    if (!PK) {
      const float4 zq = act ? *reinterpret_cast<const float4 *>(z + (it.row * kShH + q) * F + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ZLDS) zs[q][threadIdx.x] = zq; else zr[(ZLDS || PK) ? 0 : q] = zq;
    }
    acc[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  const float el_j = el[it.row * kShH + h];
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  double gl = 0.0;   // (round 6: the row's sum of signed logit-gradient terms in double, like the destination walk's)
  // (ZLDS: the registers it frees also pay for the NEXT step's ids — requested before this step's gathers, so that a step
  //  waits for one memory round trip, its gathers', instead of two dependent ones)
  ShBlock b, fp, nb, nfp;  // fp: the block's FORWARD positions (dropout), fetched beside the column ids, not behind them
  // (PK: a weight lane needs the forward positions of ITS two edges only, p0 + e and p0 + 2 + e: two dwords instead of the
  //  block's four — 4 registers that keep the dropout form at 128 = 4 wavefronts per SIMD)
  int32_t fq2[2] = {0, 0}, nfq2[2] = {0, 0};
  auto own_pos = [&](int64_t p0, int32_t (&o)[2]) {
    const int64_t a = p0 + e, c = p0 + 2 + e;
    o[0] = (a >= it.beg && a < it.end) ? posT[a] : 0;
    o[1] = (c >= it.beg && c < it.end) ? posT[c] : 0;
  };
  if (ZLDS && (it.beg & ~(int64_t)3) < it.end) {
    sh_block(col, it.beg & ~(int64_t)3, it.beg, it.end, nb);
    if (DROP && !PK) sh_block(posT, it.beg & ~(int64_t)3, it.beg, it.end, nfp);
    if (DROP && PK) own_pos(it.beg & ~(int64_t)3, nfq2);
  }
  for (int64_t p0 = it.beg & ~(int64_t)3; p0 < it.end; p0 += 4) {
    if (ZLDS) {
      b = nb;
      if (DROP && !PK) fp = nfp;
      if (DROP && PK) { fq2[0] = nfq2[0]; fq2[1] = nfq2[1]; }
      if (p0 + 4 < it.end) {
        sh_block(col, p0 + 4, it.beg, it.end, nb);
        if (DROP && !PK) sh_block(posT, p0 + 4, it.beg, it.end, nfp);
        if (DROP && PK) own_pos(p0 + 4, nfq2);
      }
    } else {
      sh_block(col, p0, it.beg, it.end, b);
      if (DROP) sh_block(posT, p0, it.beg, it.end, fp);
    }
    float4 gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) gv[u] = *reinterpret_cast<const float4 *>(gy + (int64_t)b.c[u] * F + kk);   // (idle channel lanes: kk = 0; z' = 0 there, so their dots are 0 and their T is never stored)
    const int32_t cA = e ? b.c[1] : b.c[0], cB = e ? b.c[3] : b.c[2];
    const bool okA = e ? b.ok[1] : b.ok[0], okB = e ? b.ok[3] : b.ok[2];
    const float4 st0 = *reinterpret_cast<const float4 *>(stats + ((int64_t)cA * kShH + h) * 4);
    const float4 st1 = *reinterpret_cast<const float4 *>(stats + ((int64_t)cB * kShH + h) * 4);
    float wk[2];
    if (ZLDS) asm volatile("" ::: "memory");   // (the z slots are re-read every step: hoisted out of the loop they would be the 32 registers again)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float da;
      if (PK) {
        f2v v0[4], v1[4];
        sh_pair_dots<ZLDS>(ZP, zs, sel4(e != 0, gv[2 * pr + 1], gv[2 * pr]), sel4(e != 0, gv[2 * pr], gv[2 * pr + 1]), v0, v1);
        if (ZLDS) asm volatile("" ::: "memory");
        da = sh_pair_reduce(v0, v1);
      } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < kShH; ++q) {
          const float4 zq = ZLDS ? zs[q][threadIdx.x] : zr[(ZLDS || PK) ? 0 : q];
          v[q] = dot4(zq, gv[2 * pr]);
          v[8 + q] = dot4(zq, gv[2 * pr + 1]);
        }
        if (ZLDS) asm volatile("" ::: "memory");
        da = reduce_scatter16(v, li);
      }
      const float4 st = pr ? st1 : st0;
      const bool ok = pr ? okB : okA;
      const float raw = el_j + st.x;
      const float al = fexp(lrelu(raw, d.slope) - st.y) * st.z;
      float alk = al;
      if (DROP) {  // the keep bit lives at the FORWARD position of the edge
        const int32_t fq = (PK && ZLDS) ? fq2[pr] : (pr ? (e ? fp.c[3] : fp.c[2]) : (e ? fp.c[1] : fp.c[0]));
        const bool keep = ok && drop_word((int64_t)fq, kShH, h, offset, seed) >= d.drop_thresh;
        alk = keep ? al * d.drop_scale : 0.0f;
        da = keep ? da * d.drop_scale : 0.0f;
      }
      const float ds = al * (da - st.w);
      const float dv = raw > 0.0f ? ds : ds * d.slope;
      gl += (double)(ok ? dv : 0.0f);
      wk[pr] = ok ? alk : 0.0f;
    }
    sh_accumulate<0>(wk[0], gv[0], acc);
    sh_accumulate<1>(wk[0], gv[1], acc);
    sh_accumulate<0>(wk[1], gv[2], acc);
    sh_accumulate<1>(wk[1], gv[3], acc);
  }
  gl += row_ror8_f64(gl);
  if (it.is_chunk) {
    if (act) {
#pragma unroll
      for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(pacc + (it.cid * kShH + q) * F + kk) = acc[q];
    }
    if (e == 0) pgel[it.cid * kShH + h] = (float)gl;
    return;
  }
  if (act) {
#pragma unroll
    for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(T + (it.row * kShH + q) * F + kk) = acc[q];
  }
  if (e == 0) gel[it.row * kShH + h] = (float)gl;
}

// PIPE (round 6, option gat_sh_pipe, A/B): the source walk above with its GATHERS software-pipelined — the gy rows and the stats panels of
// step i + 1 are requested before step i's dots (their ids two steps ahead), so a wavefront's own arithmetic hides its own memory round trip
// instead of relying on the 2-3 other wavefronts of the SIMD; costs ~24 more registers (the in-flight step's rows).  Packed pair dots, z_j pairs
// in LDS slots (the PK + ZLDS form); same arithmetic in the same order: same bits.
template <bool DROP>
__global__ __launch_bounds__(kBlock) void gat_sh_bwd_src_pipe_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col /* colT */, const int32_t *__restrict__ posT,
    const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ chunk_ptr, const float *__restrict__ el, const float *__restrict__ z,
    const float *__restrict__ gy, const float *__restrict__ stats, float *__restrict__ T, float *__restrict__ gel,
    float *__restrict__ pacc, float *__restrict__ pgel, const int64_t *__restrict__ rng, const ShDims d) {
  GGL_SH_PROLOGUE();
  const int64_t F = d.F;
  __shared__ float4 zs[kShH][kBlock];
  float4 acc[kShH];
  ShPairs ZP;
  sh_pairs_load(ZP, z + it.row * kShH * F + kk, F, li, act);
  sh_pairs_to_lds(ZP, zs);
#pragma unroll
  for (int q = 0; q < kShH; ++q) acc[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const float el_j = el[it.row * kShH + h];
  const uint64_t seed = DROP ? (uint64_t)rng[0] : 0, offset = DROP ? (uint64_t)rng[1] : 0;
  double gl = 0.0;
  const int64_t pfirst = it.beg & ~(int64_t)3;
  if (pfirst >= it.end) {   // an empty row: zeros
    if (!it.is_chunk) {
      if (act) {
#pragma unroll
        for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(T + (it.row * kShH + q) * F + kk) = acc[q];
      }
      if (e == 0) gel[it.row * kShH + h] = 0.0f;
    } else {
      if (act) {
#pragma unroll
        for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(pacc + (it.cid * kShH + q) * F + kk) = acc[q];
      }
      if (e == 0) pgel[it.cid * kShH + h] = 0.0f;
    }
    return;
  }
  auto own_pos = [&](int64_t p0, int32_t (&o)[2]) {
    const int64_t a = p0 + e, c = p0 + 2 + e;
    o[0] = (a >= it.beg && a < it.end) ? posT[a] : 0;
    o[1] = (c >= it.beg && c < it.end) ? posT[c] : 0;
  };
  // in flight: the rows / stats of the step about to be computed (N = "next"), and the ids of the step after it (nb)
  ShBlock bN, nb;
  int32_t fqN[2] = {0, 0}, nfq[2] = {0, 0};
  float4 gvN[4], st0N, st1N;
  auto request = [&](const ShBlock &b) {
#pragma unroll
    for (int u = 0; u < 4; ++u) gvN[u] = *reinterpret_cast<const float4 *>(gy + (int64_t)b.c[u] * F + kk);
    const int32_t cA = e ? b.c[1] : b.c[0], cB = e ? b.c[3] : b.c[2];
    st0N = *reinterpret_cast<const float4 *>(stats + ((int64_t)cA * kShH + h) * 4);
    st1N = *reinterpret_cast<const float4 *>(stats + ((int64_t)cB * kShH + h) * 4);
  };
  sh_block(col, pfirst, it.beg, it.end, bN);
  if (DROP) own_pos(pfirst, fqN);
  request(bN);
  if (pfirst + 4 < it.end) {
    sh_block(col, pfirst + 4, it.beg, it.end, nb);
    if (DROP) own_pos(pfirst + 4, nfq);
  }
  for (int64_t p0 = pfirst; p0 < it.end; p0 += 4) {
    // this step's operands leave the in-flight set ...
    const ShBlock b = bN;
    const int32_t fq2[2] = {fqN[0], fqN[1]};
    float4 gv[4] = {gvN[0], gvN[1], gvN[2], gvN[3]};
    const float4 st0 = st0N, st1 = st1N;
    // ... and the next step's are requested before this step's arithmetic
    if (p0 + 4 < it.end) {
      bN = nb;
      if (DROP) { fqN[0] = nfq[0]; fqN[1] = nfq[1]; }
      request(bN);
      if (p0 + 8 < it.end) {
        sh_block(col, p0 + 8, it.beg, it.end, nb);
        if (DROP) own_pos(p0 + 8, nfq);
      }
    }
    const bool okA = e ? b.ok[1] : b.ok[0], okB = e ? b.ok[3] : b.ok[2];
    float wk[2];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f2v v0[4], v1[4];
      sh_pair_dots<true>(ZP, zs, sel4(e != 0, gv[2 * pr + 1], gv[2 * pr]), sel4(e != 0, gv[2 * pr], gv[2 * pr + 1]), v0, v1);
      asm volatile("" ::: "memory");
      float da = sh_pair_reduce(v0, v1);
      const float4 st = pr ? st1 : st0;
      const bool ok = pr ? okB : okA;
      const float raw = el_j + st.x;
      const float al = fexp(lrelu(raw, d.slope) - st.y) * st.z;
      float alk = al;
      if (DROP) {
        const bool keep = ok && drop_word((int64_t)fq2[pr], kShH, h, offset, seed) >= d.drop_thresh;
        alk = keep ? al * d.drop_scale : 0.0f;
        da = keep ? da * d.drop_scale : 0.0f;
      }
      const float ds = al * (da - st.w);
      const float dv = raw > 0.0f ? ds : ds * d.slope;
      gl += (double)(ok ? dv : 0.0f);
      wk[pr] = ok ? alk : 0.0f;
    }
    sh_accumulate<0>(wk[0], gv[0], acc);
    sh_accumulate<1>(wk[0], gv[1], acc);
    sh_accumulate<0>(wk[1], gv[2], acc);
    sh_accumulate<1>(wk[1], gv[3], acc);
  }
  gl += row_ror8_f64(gl);
  if (it.is_chunk) {
    if (act) {
#pragma unroll
      for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(pacc + (it.cid * kShH + q) * F + kk) = acc[q];
    }
    if (e == 0) pgel[it.cid * kShH + h] = (float)gl;
    return;
  }
  if (act) {
#pragma unroll
    for (int q = 0; q < kShH; ++q) *reinterpret_cast<float4 *>(T + (it.row * kShH + q) * F + kk) = acc[q];
  }
  if (e == 0) gel[it.row * kShH + h] = (float)gl;
}

}  // namespace ggl

using namespace ggl;

// ---- fast path entry points (see the block comment above gat_fwd2_kernel) ---------------------------------
static int gat_fast_c4(int64_t H, int64_t C) {
  if (H <= 0 || C <= 0 || C % 4 != 0) return 0;
  const int64_t c4 = C / 4;
  if (c4 != 1 && c4 != 2 && c4 != 4 && c4 != 8 && c4 != 16) return 0;
  if (H * c4 > 64) return 0;
  return (int)c4;
}

extern "C" int ggl_gat_fast_supported(int64_t H, int64_t C) { return gat_fast_c4(H, C) ? 1 : 0; }

static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// dispatch KERN<C4, DROP, OFF32> over the supported head widths
#define GGL_GAT2_DISPATCH(KERN, c4, drop, off32, ...)                                                  \
  do {                                                                                                 \
    const int sel_ = ((drop) ? 2 : 0) | ((off32) ? 1 : 0);                                             \
    switch ((c4) * 4 + sel_) {                                                                         \
      GGL_GAT2_CASES(KERN, 1, __VA_ARGS__) GGL_GAT2_CASES(KERN, 2, __VA_ARGS__)                        \
      GGL_GAT2_CASES(KERN, 4, __VA_ARGS__) GGL_GAT2_CASES(KERN, 8, __VA_ARGS__)                        \
      GGL_GAT2_CASES(KERN, 16, __VA_ARGS__)                                                            \
      default: break;                                                                                  \
    }                                                                                                  \
  } while (0)
#define GGL_GAT2_CASES(KERN, C4, ...)                                                                  \
  case C4 * 4 + 0: GGL_LAUNCH((KERN<C4, false, false>), grid, kBlock, s, __VA_ARGS__); break;          \
  case C4 * 4 + 1: GGL_LAUNCH((KERN<C4, false, true>), grid, kBlock, s, __VA_ARGS__); break;           \
  case C4 * 4 + 2: GGL_LAUNCH((KERN<C4, true, false>), grid, kBlock, s, __VA_ARGS__); break;           \
  case C4 * 4 + 3: GGL_LAUNCH((KERN<C4, true, true>), grid, kBlock, s, __VA_ARGS__); break;

extern "C" int ggl_gat_fast_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el,
                                const float *er, const float *x, int64_t N_src, float slope, int64_t H,
                                int64_t C, float p_drop, int64_t *rng_state, float *out, float *rowmax,
                                float *rowden, void *stream) {
  GGL_REQUIRE(plan && plan->rowptr, GGL_EINVAL, "plan is NULL");
  const int c4 = gat_fast_c4(H, C);
  GGL_REQUIRE(c4 > 0 && plan->chunk > 0, GGL_EINVAL, "head shape not supported by the fast GAT path");
  const int64_t N = plan->N;
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(er && out && rowmax && rowden, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((col && el && x) || plan->E == 0, GGL_EINVAL, "NULL pointer");
  GatDims d{};
  d.slope = slope; d.N = N; d.H = H; d.C = C; d.K = H * C; d.E = plan->E;
  d.chunk = plan->chunk; d.n_long = plan->n_long; d.n_chunks = plan->n_long > 0 ? plan->n_chunks : 0;
  int rcd = set_dropout(d, p_drop, rng_state);
  if (rcd) return rcd;
  float *pacc = nullptr, *pm = nullptr, *pd = nullptr;
  if (plan->n_long > 0) {
    GGL_REQUIRE(plan->long_rows && plan->chunk_ptr && plan->partial, GGL_EWORKSPACE,
                "plan has long rows but long_rows/chunk_ptr/partial is NULL");
    pacc = static_cast<float *>(plan->partial);
    pm = pacc + plan->n_chunks * d.K;
    pd = pm + plan->n_chunks * H;
  }
  GGL_REQUIRE(al16(x) && al16(out) && al16(pacc) && al16(col), GGL_EINVAL, "fast GAT path needs 16-byte aligned buffers");
  d.logL = pow2_log2(H * c4);
  const int64_t items = d.n_chunks + N;
  const int64_t grid = ceil_div(items << d.logL, (int64_t)kBlock);
  GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  const int32_t *order = options().row_order ? plan->row_order : nullptr;
  const bool off32 = N_src > 0 && N_src * d.K * 4 < ((int64_t)1 << 32);
  hipStream_t s = as_stream(stream);
  GGL_GAT2_DISPATCH(gat_fwd2_kernel, c4, d.drop_thresh != 0, off32, plan->rowptr, col, order, plan->long_rows,
                    plan->chunk_ptr, el, er, x, out, rowmax, rowden, pacc, pm, pd, (const int64_t *)rng_state, d);
  GGL_LAUNCH_CHECK();
  if (plan->n_long > 0) {
    GGL_LAUNCH((gat_long_final_kernel), plan->n_long, kBlock, s, plan->long_rows, plan->chunk_ptr,
               (const float *)pacc, (const float *)pm, (const float *)pd, out, rowmax, rowden, d);
    GGL_LAUNCH_CHECK();
  }
  if (d.drop_thresh) return rng_advance(rng_state, stream);
  return GGL_OK;
}

// Both walks of the backward.  stats: workspace of N_dst * H * 4 floats (16-byte aligned).  plan->partial:
// >= n_chunks * H floats; planT->partial: ggl_partial_bytes(GGL_F32, n_chunksT, H*C + H, 0) bytes.
extern "C" int ggl_gat_fast_bwd(const ggl_segplan_t *plan, const int32_t *col, const ggl_segplan_t *planT,
                                const int32_t *colT, const int32_t *posT, const float *el, const float *er,
                                const float *x, const float *g, const float *out, const float *rowmax,
                                const float *rowden, float slope, int64_t H, int64_t C, float p_drop,
                                const int64_t *rng_used, float *stats, float *gx, float *gel, float *ger,
                                void *stream) {
  GGL_REQUIRE(plan && plan->rowptr && planT && planT->rowptr, GGL_EINVAL, "plan is NULL");
  const int c4 = gat_fast_c4(H, C);
  GGL_REQUIRE(c4 > 0 && plan->chunk > 0 && planT->chunk > 0, GGL_EINVAL, "head shape not supported by the fast GAT path");
  const int64_t N = plan->N, NT = planT->N, E = plan->E;
  GGL_REQUIRE(planT->E == E, GGL_EINVAL, "forward and transposed plans disagree");
  hipStream_t s = as_stream(stream);
  GatDims d{};
  d.slope = slope; d.H = H; d.C = C; d.K = H * C; d.E = E;
  int rcd = set_dropout(d, p_drop, rng_used);
  if (rcd) return rcd;
  GGL_REQUIRE(p_drop == 0.0f || posT || E == 0, GGL_EINVAL, "attention dropout needs posT");
  d.logL = pow2_log2(H * c4);
  const bool off32 = (N > NT ? N : NT) * d.K * 4 < ((int64_t)1 << 32);
  if (N > 0) {  // destination walk
    GGL_REQUIRE(er && g && out && rowmax && rowden && ger && stats, GGL_EINVAL, "NULL pointer");
    GGL_REQUIRE((col && el && x) || E == 0, GGL_EINVAL, "NULL pointer");
    d.N = N; d.chunk = plan->chunk; d.n_long = plan->n_long; d.n_chunks = plan->n_long > 0 ? plan->n_chunks : 0;
    float *pger = nullptr;
    if (plan->n_long > 0) {
      GGL_REQUIRE(plan->long_rows && plan->chunk_ptr && plan->partial, GGL_EWORKSPACE,
                  "plan has long rows but long_rows/chunk_ptr/partial is NULL");
      pger = static_cast<float *>(plan->partial);
    }
    GGL_REQUIRE(al16(x) && al16(g) && al16(out) && al16(stats) && al16(col), GGL_EINVAL,
                "fast GAT path needs 16-byte aligned buffers");
    const int64_t grid = ceil_div((d.n_chunks + N) << d.logL, (int64_t)kBlock);
    GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
    const int32_t *order = options().row_order ? plan->row_order : nullptr;
    GGL_GAT2_DISPATCH(gat_bwd_dst2_kernel, c4, d.drop_thresh != 0, off32, plan->rowptr, col, order,
                      plan->long_rows, plan->chunk_ptr, el, er, x, g, out, rowmax, rowden, stats, ger, pger,
                      rng_used, d);
    GGL_LAUNCH_CHECK();
    if (plan->n_long > 0) {
      GGL_LAUNCH((gat_bwd_dst_final4_kernel), gat_grid_for(plan->n_long * H), kBlock, s, plan->long_rows,
                 plan->chunk_ptr, reinterpret_cast<const double *>(pger), ger, plan->n_long, H);
      GGL_LAUNCH_CHECK();
    }
  }
  if (NT > 0) {  // source walk
    GGL_REQUIRE(gx && gel && el && x, GGL_EINVAL, "NULL pointer");
    GGL_REQUIRE((colT && g && stats) || E == 0, GGL_EINVAL, "NULL pointer");
    d.N = NT; d.chunk = planT->chunk; d.n_long = planT->n_long; d.n_chunks = planT->n_long > 0 ? planT->n_chunks : 0;
    float *pacc = nullptr, *pgel = nullptr;
    if (planT->n_long > 0) {
      GGL_REQUIRE(planT->long_rows && planT->chunk_ptr && planT->partial, GGL_EWORKSPACE,
                  "transposed plan has long rows but long_rows/chunk_ptr/partial is NULL");
      pacc = static_cast<float *>(planT->partial);
      pgel = pacc + planT->n_chunks * d.K;
    }
    GGL_REQUIRE(al16(x) && al16(g) && al16(gx) && al16(pacc) && al16(colT), GGL_EINVAL,
                "fast GAT path needs 16-byte aligned buffers");
    const int64_t grid = ceil_div((d.n_chunks + NT) << d.logL, (int64_t)kBlock);
    GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
    const int32_t *order = options().row_order ? planT->row_order : nullptr;
    GGL_GAT2_DISPATCH(gat_bwd_src2_kernel, c4, d.drop_thresh != 0, off32, planT->rowptr, colT, posT, order,
                      planT->long_rows, planT->chunk_ptr, el, x, g, (const float *)stats, gx, gel, pacc, pgel,
                      rng_used, d);
    GGL_LAUNCH_CHECK();
    if (planT->n_long > 0) {
      GGL_LAUNCH((gat_bwd_src_final_kernel), planT->n_long, kBlock, s, planT->long_rows, planT->chunk_ptr,
                 (const float *)pacc, (const float *)pgel, gx, gel, d);
      GGL_LAUNCH_CHECK();
    }
  }
  return GGL_OK;
}

// ---- head-mean GAT with a shared input row: entry points (see the block comment above gat_sh_rowmax_kernel) ----
extern "C" int ggl_gat_sh_supported(int64_t H, int64_t F, int64_t C) {
  return (H == 8 && F > 0 && F <= 64 && F % 4 == 0 && C > 0 && C <= 64) ? 1 : 0;
}

// floats of plan->partial the forward needs for a plan with n_chunks hub chunks (the backward needs 8 per chunk
// on the forward plan and 8 * Cp + 8 per chunk of the transposed plan)
extern "C" size_t ggl_gat_sh_partial_bytes(int64_t n_chunks, int64_t F) {
  if (n_chunks <= 0) return 0;
  return (size_t)n_chunks * (size_t)(8 * F + 16) * sizeof(float) + 64;
}

static int sh_dims(ShDims &d, const ggl_segplan_t *plan, int64_t F, float slope, float p_drop, const int64_t *rng) {
  GGL_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "p_drop must be in [0, 1)");
  GGL_REQUIRE(p_drop == 0.0f || rng, GGL_EINVAL, "attention dropout needs an rng_state");
  d.slope = slope; d.N = plan->N; d.F = F; d.E = plan->E;
  d.chunk = plan->chunk; d.n_long = plan->n_long; d.n_chunks = plan->n_long > 0 ? plan->n_chunks : 0;
  d.drop_thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  d.drop_scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  if (plan->n_long > 0)
    GGL_REQUIRE(plan->long_rows && plan->chunk_ptr && plan->partial, GGL_EWORKSPACE,
                "plan has long rows but long_rows/chunk_ptr/partial is NULL");
  return GGL_OK;
}

// A[N,8,F] = sum_j alpha_ijh x[j,:] (normalised), den[N,8], rowmax[N,8]; x[N_src,F], el[N_src,8], er[N,8]
extern "C" int ggl_gat_sh_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el, const float *er,
                              const float *x, int64_t F, float slope, float p_drop, int64_t *rng_state,
                              float *rowmax, float *A, float *den, void *stream) {
  GGL_REQUIRE(plan && plan->rowptr && plan->chunk > 0, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(ggl_gat_sh_supported(8, F, 1), GGL_EINVAL, "row width not supported by the shared-row GAT path");
  if (plan->N == 0) return GGL_OK;
  GGL_REQUIRE(er && rowmax && A && den, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((col && el && x) || plan->E == 0, GGL_EINVAL, "NULL pointer");
  ShDims d{};
  int rc = sh_dims(d, plan, F, slope, p_drop, rng_state);
  if (rc) return rc;
  float *pacc = nullptr, *pden = nullptr, *pmax = nullptr;
  if (plan->n_long > 0) {
    pacc = static_cast<float *>(plan->partial);
    pden = pacc + plan->n_chunks * 8 * F;
    pmax = pden + plan->n_chunks * 8;
  }
  GGL_REQUIRE(al16(x) && al16(A) && al16(pacc) && al16(col), GGL_EINVAL, "shared-row GAT path needs 16-byte aligned buffers");
  hipStream_t s = as_stream(stream);
  const int64_t grid = ceil_div((d.n_chunks + d.N) * 16, (int64_t)kBlock);
  GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  const int32_t *order = options().row_order ? plan->row_order : nullptr;
  GGL_LAUNCH((gat_sh_rowmax_kernel), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows, plan->chunk_ptr, el,
             er, rowmax, pmax, d);
  GGL_LAUNCH_CHECK();
  if (plan->n_long > 0) {
    GGL_LAUNCH((gat_sh_rowmax_final_kernel), gat_grid_for(plan->n_long * 8), kBlock, s, plan->long_rows, plan->chunk_ptr,
               (const float *)pmax, rowmax, plan->n_long);
    GGL_LAUNCH_CHECK();
  }
  const bool pf = options().gat_sh_prefetch != 0;
  if (d.drop_thresh && pf)
    GGL_LAUNCH((gat_sh_fwd_kernel<true, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows, plan->chunk_ptr,
               el, er, (const float *)rowmax, x, A, den, pacc, pden, (const int64_t *)rng_state, d);
  else if (d.drop_thresh)
    GGL_LAUNCH((gat_sh_fwd_kernel<true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows, plan->chunk_ptr,
               el, er, (const float *)rowmax, x, A, den, pacc, pden, (const int64_t *)rng_state, d);
  else if (pf)
    GGL_LAUNCH((gat_sh_fwd_kernel<false, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows, plan->chunk_ptr,
               el, er, (const float *)rowmax, x, A, den, pacc, pden, (const int64_t *)rng_state, d);
  else
    GGL_LAUNCH((gat_sh_fwd_kernel<false>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows, plan->chunk_ptr,
               el, er, (const float *)rowmax, x, A, den, pacc, pden, (const int64_t *)rng_state, d);
  GGL_LAUNCH_CHECK();
  if (plan->n_long > 0) {
    GGL_LAUNCH((gat_sh_fwd_final_kernel), plan->n_long, kBlock, s, plan->long_rows, plan->chunk_ptr, (const float *)pacc,
               (const float *)pden, A, den, plan->n_long, F);
    GGL_LAUNCH_CHECK();
  }
  if (d.drop_thresh) return rng_advance(rng_state, stream);
  return GGL_OK;
}

// destination walk (ger) then source walk (T, gel).  G[N,8,F] = dL/dA, stats[N,8,4] = {er, m, 1/(den+1e-16),
// <G_ih, A_ih>}, z[N_src,8,Cp] = the rows' own transformed features, gy[N,Cp] = the per-row output gradient the
// head mean spreads over the heads (dL/dA_ih = gy_i W_h^T), Cp <= 64 a multiple of 4.
// stats[i,h] = {er, rowmax, 1 / (den + 1e-16), <G_ih, A_ih>} for the head-mean backward in ONE pass over G and A (round 5):
// the hosts built it from five torch launches — an [N, 8, F] product, a reduce over its innermost 64 floats that alone took
// 1.2 ms on the Reddit-sized graph (240 MB that stream in 0.05 ms), a reciprocal, an add and a stack: 2 ms of a 28 ms step.
// 16 lanes (one DPP row) per (row, head): a float4 each per 64 columns, summed over the row with four xor-shuffles.
__global__ __launch_bounds__(kBlock) void gat_sh_stats_kernel(const float *__restrict__ er, const float *__restrict__ rowmax,
                                                              const float *__restrict__ den, const float *__restrict__ G,
                                                              const float *__restrict__ A, int64_t NH, int64_t F,
                                                              float *__restrict__ stats) {
  const int li = (int)threadIdx.x & 15;
  const int64_t stride = grid_threads() >> 4;
  for (int64_t r = thread_id() >> 4; r < NH; r += stride) {
    float acc = 0.0f;
    for (int64_t k = 4 * li; k < F; k += 64) {     // (F % 4 == 0: ggl_gat_sh_supported)
      const float4 g = *reinterpret_cast<const float4 *>(G + r * F + k), a = *reinterpret_cast<const float4 *>(A + r * F + k);
      acc += dot4(g, a);
    }
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (li == 0) *reinterpret_cast<float4 *>(stats + r * 4) = make_float4(er[r], rowmax[r], 1.0f / (den[r] + 1e-16f), acc);
  }
}

extern "C" int ggl_gat_sh_stats(const float *er, const float *rowmax, const float *den, const float *G, const float *A,
                                int64_t N, int64_t F, float *stats, void *stream) {
  GGL_REQUIRE(N >= 0 && F > 0 && F % 4 == 0, GGL_EINVAL, "ggl_gat_sh_stats: F must be a positive multiple of 4");
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(er && rowmax && den && G && A && stats, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(al16(G) && al16(A) && al16(stats), GGL_EINVAL, "ggl_gat_sh_stats needs 16-byte aligned buffers");
  const int64_t NH = N * kShH;
  int64_t grid = ceil_div(NH * 16, (int64_t)kBlock);
  if (grid > 65536) grid = 65536;
  GGL_LAUNCH((gat_sh_stats_kernel), grid, kBlock, as_stream(stream), er, rowmax, den, G, A, NH, F, stats);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_gat_sh_bwd(const ggl_segplan_t *plan, const int32_t *col, const ggl_segplan_t *planT,
                              const int32_t *colT, const int32_t *posT, const float *el, const float *x, int64_t F,
                              const float *G, const float *stats, const float *z, const float *gy, int64_t Cp,
                              float slope, float p_drop, const int64_t *rng_used, float *ger, float *T, float *gel,
                              void *stream) {
  GGL_REQUIRE(plan && plan->rowptr && planT && planT->rowptr && plan->chunk > 0 && planT->chunk > 0, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(ggl_gat_sh_supported(8, F, Cp) && Cp % 4 == 0, GGL_EINVAL, "shape not supported by the shared-row GAT path");
  GGL_REQUIRE(planT->E == plan->E, GGL_EINVAL, "forward and transposed plans disagree");
  GGL_REQUIRE(p_drop == 0.0f || posT || plan->E == 0, GGL_EINVAL, "attention dropout needs posT");
  hipStream_t s = as_stream(stream);
  if (plan->N > 0) {
    GGL_REQUIRE(ger && G && stats && el && x, GGL_EINVAL, "NULL pointer");
    ShDims d{};
    int rc = sh_dims(d, plan, F, slope, p_drop, rng_used);
    if (rc) return rc;
    // (forward plan's partial: FOUR doubles per chunk and head since round 6 = 256 bytes per chunk: ggl_gat_sh_partial_bytes(n_chunks, 8))
    float *pger = plan->n_long > 0 ? static_cast<float *>(plan->partial) : nullptr;
    GGL_REQUIRE(al16(x) && al16(G) && al16(stats) && al16(col) && al16(pger), GGL_EINVAL, "shared-row GAT path needs 16-byte aligned buffers");
    const int64_t grid = ceil_div((d.n_chunks + d.N) * 16, (int64_t)kBlock);
    GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
    const int32_t *order = options().row_order ? plan->row_order : nullptr;
    const bool pf = options().gat_sh_prefetch != 0;
    if (pf && options().gat_sh_pk != 0 && options().gat_sh_glds == 0) {   // round 6 default: packed pair dots, select-free reduce
      if (d.drop_thresh)
        GGL_LAUNCH((gat_sh_bwd_dst_kernel<true, true, false, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                   plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
      else
        GGL_LAUNCH((gat_sh_bwd_dst_kernel<false, true, false, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                   plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
    } else if (pf && options().gat_sh_glds != 0) {
      if (d.drop_thresh)
        GGL_LAUNCH((gat_sh_bwd_dst_kernel<true, true, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                   plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
      else
        GGL_LAUNCH((gat_sh_bwd_dst_kernel<false, true, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                   plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
    } else if (d.drop_thresh && pf)
      GGL_LAUNCH((gat_sh_bwd_dst_kernel<true, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                 plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
    else if (d.drop_thresh)
      GGL_LAUNCH((gat_sh_bwd_dst_kernel<true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                 plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
    else if (pf)
      GGL_LAUNCH((gat_sh_bwd_dst_kernel<false, true>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                 plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
    else
      GGL_LAUNCH((gat_sh_bwd_dst_kernel<false>), grid, kBlock, s, plan->rowptr, col, order, plan->long_rows,
                 plan->chunk_ptr, el, x, G, stats, ger, pger, rng_used, d);
    GGL_LAUNCH_CHECK();
    if (plan->n_long > 0) {
      GGL_LAUNCH((gat_bwd_dst_final4_kernel), gat_grid_for(plan->n_long * 8), kBlock, s, plan->long_rows, plan->chunk_ptr,
                 reinterpret_cast<const double *>(pger), ger, plan->n_long, (int64_t)8);
      GGL_LAUNCH_CHECK();
    }
  }
  if (planT->N > 0) {
    GGL_REQUIRE(T && gel && z && gy && stats && el, GGL_EINVAL, "NULL pointer");
    ShDims d{};
    int rc = sh_dims(d, planT, Cp, slope, p_drop, rng_used);
    if (rc) return rc;
    float *pacc = nullptr, *pgel = nullptr;
    if (planT->n_long > 0) {
      pacc = static_cast<float *>(planT->partial);
      pgel = pacc + planT->n_chunks * 8 * Cp;
    }
    GGL_REQUIRE(al16(z) && al16(gy) && al16(T) && al16(pacc) && al16(colT), GGL_EINVAL, "shared-row GAT path needs 16-byte aligned buffers");
    const int64_t grid = ceil_div((d.n_chunks + d.N) * 16, (int64_t)kBlock);
    GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
    const int32_t *order = options().row_order ? planT->row_order : nullptr;
    const bool pk = options().gat_sh_pk != 0 && options().gat_sh_zlds != 0;
    if (options().gat_sh_pk != 0 && options().gat_sh_zlds == 0) {   // A/B: packed pair dots with the row's z_j pairs in REGISTERS (no LDS slots)
      if (d.drop_thresh)
        GGL_LAUNCH((gat_sh_bwd_src_kernel<true, 1, false, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                   planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
      else
        GGL_LAUNCH((gat_sh_bwd_src_kernel<false, 1, false, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                   planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    } else if (pk && options().gat_sh_pipe != 0) {    // A/B: gathers software-pipelined one step ahead
      if (d.drop_thresh)
        GGL_LAUNCH((gat_sh_bwd_src_pipe_kernel<true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                   planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
      else
        GGL_LAUNCH((gat_sh_bwd_src_pipe_kernel<false>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                   planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    } else if (pk && d.drop_thresh && options().gat_sh_waves >= 4)   // (A/B: built for 4 wavefronts per SIMD — 128 registers + 12 spilled values)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<true, 4, true, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else if (pk && d.drop_thresh)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<true, 1, true, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else if (pk)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<false, 1, true, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else if (d.drop_thresh && options().gat_sh_zlds != 0)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<true, 1, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else if (d.drop_thresh && options().gat_sh_waves >= 4)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<true, 4>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else if (d.drop_thresh)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else if (options().gat_sh_zlds != 0)
      GGL_LAUNCH((gat_sh_bwd_src_kernel<false, 1, true>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    else
      GGL_LAUNCH((gat_sh_bwd_src_kernel<false>), grid, kBlock, s, planT->rowptr, colT, posT, order, planT->long_rows,
                 planT->chunk_ptr, el, z, gy, stats, T, gel, pacc, pgel, rng_used, d);
    GGL_LAUNCH_CHECK();
    if (planT->n_long > 0) {
      GatDims gd{};
      gd.H = 8; gd.K = 8 * Cp; gd.n_long = planT->n_long;
      GGL_LAUNCH((gat_bwd_src_final_kernel), planT->n_long, kBlock, s, planT->long_rows, planT->chunk_ptr,
                 (const float *)pacc, (const float *)pgel, T, gel, gd);
      GGL_LAUNCH_CHECK();
    }
  }
  return GGL_OK;
}

