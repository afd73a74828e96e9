// gammagl_amd/csrc/backward.hip — edge-parallel kernels: the backward passes of the segment ops and
// bspmm's per-edge weight gradient.  These have no reduction across edges, so they are plain
// coalesced row copies / per-edge dot products.
//   ggl_segment_sum_bwd  : SegmentSum::backward  (src/segment_sum.cpp:43-54)   gin[e] = gout[ids[e]]
//   ggl_segment_mean_bwd : SegmentMean::backward (src/segment_mean.cpp:44-63)  ... / bincount[ids[e]]
//   ggl_segment_max_bwd  : SegmentMax::backward  (src/segment_max.cpp:48-61)   scatter by argmax
//   ggl_bspmm_grad_w     : bspmm_sum_cpu_backward's grad_weight (cpu/bspmm_sum_cpu.cpp:102-107)
#include "common.hpp"

namespace ggl {

// One lane group of L = 2^logL lanes per destination row e; VEC elements per lane per step.
// MEAN divides by the segment's element count (int64 -> storage float type, then a rounded divide).
template <typename T, int VEC, bool MEAN>
__global__ __launch_bounds__(kBlock) void row_gather_kernel(const typename TT<T>::S *src,
                                                            const int64_t *ids,
                                                            const int64_t *rowptr, int64_t E,
                                                            int64_t K, int logL,
                                                            typename TT<T>::S *dst) {
  using S = typename TT<T>::S;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int L = 1 << logL;
  const int64_t e = (block_id() * kWavesPerBlock + wave) * (kWave >> logL) + (lane >> logL);
  if (e >= E) return;
  const int li = lane & (L - 1);
  const int64_t s = ids[e];
  typename TT<T>::A cnt = TT<T>::zero();
  // torch promotes the int64 bincount to the gradient dtype before dividing (segment_mean.cpp:61)
  if (MEAN) cnt = TT<T>::load(TT<T>::store((typename TT<T>::A)(rowptr[s + 1] - rowptr[s])));
  for (int64_t kk = (int64_t)li * VEC; kk < K; kk += (int64_t)L * VEC) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (kk + i < K) {
        S v = src[s * K + kk + i];
        if (MEAN) v = TT<T>::store(TT<T>::div(TT<T>::load(v), cnt));
        dst[e * K + kk + i] = v;
      }
    }
  }
}

// gin pre-zeroed; one thread per (s,k): gin[arg[s,k], k] = gout[s,k] when 0 <= arg < E
template <typename S>
__global__ __launch_bounds__(kBlock) void max_scatter_kernel(const S *gout, const int64_t *arg,
                                                             int64_t total, int64_t K, int64_t E,
                                                             S *gin) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < total; i += stride) {
    const int64_t a = arg[i];
    if (a >= 0 && a < E) gin[a * K + (i % K)] = gout[i];
  }
}

// gw[e,h] = sum_c x[src,h,c] * g[dst,h,c]   (serial over c, rounded multiply then rounded add)
__global__ __launch_bounds__(kBlock) void bspmm_grad_w_kernel(const int64_t *index, const float *x,
                                                              const float *g, int64_t E, int64_t H,
                                                              int64_t C, float *gw) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < E * H; i += stride) {
    const int64_t e = i / H, h = i - e * H;
    const float *xr = x + (index[e] * H + h) * C;
    const float *gr = g + (index[e + E] * H + h) * C;
    float acc = 0.0f;
    if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(xr) | reinterpret_cast<uintptr_t>(gr)) & 15u) == 0) {
      for (int64_t c = 0; c < C; c += 4) {  // 16-byte loads; same products added in the same order
        const float4 a = *reinterpret_cast<const float4 *>(xr + c);
        const float4 b = *reinterpret_cast<const float4 *>(gr + c);
        acc = __fadd_rn(acc, __fmul_rn(a.x, b.x));
        acc = __fadd_rn(acc, __fmul_rn(a.y, b.y));
        acc = __fadd_rn(acc, __fmul_rn(a.z, b.z));
        acc = __fadd_rn(acc, __fmul_rn(a.w, b.w));
      }
    } else {
      for (int64_t c = 0; c < C; ++c) acc = __fadd_rn(acc, __fmul_rn(xr[c], gr[c]));
    }
    gw[i] = acc;
  }
}

#ifndef GGL_EMULATE
// Wide heads (C > 16), GPU build only: a group of 2^LOGG lanes per (edge, head) splits the channels — each
// strip x[src,h,:] / g[dst,h,:] is one coalesced read instead of a thread's C/4 strided ones.  The dot is then
// summed ACROSS the lanes in channel order: the running sum ripples from lane to lane through wave shuffles, each
// lane adding its four products in order, so the result is the reference's serial sum over c bit for bit
// (bspmm_sum_cpu.cpp:95-107; pinned by the golden weight gradients).  kPerGroup consecutive items per group keep
// their loads in flight together; the last lane of the ripple holds the dot and writes it.
constexpr int kGradWPerGroup = 4;

template <int LOGG>
__global__ __launch_bounds__(kBlock) void bspmm_grad_w_wide_kernel(const int64_t *__restrict__ index,
                                                                   const float *__restrict__ x,
                                                                   const float *__restrict__ g, int64_t E,
                                                                   int64_t H, int64_t C,
                                                                   float *__restrict__ gw) {
  constexpr int G = 1 << LOGG, U = kGradWPerGroup;
  const int64_t total = E * H;
  const int64_t base = (thread_id() >> LOGG) * U;
  const int sub = (int)(threadIdx.x & (G - 1));
  if (base >= total) return;  // whole groups leave together
  const float *xr[U], *gr[U];
  float part[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t it = base + u < total ? base + u : base;
    const int64_t e = it / H, h = it - e * H;
    xr[u] = x + (index[e] * H + h) * C;
    gr[u] = g + (index[e + E] * H + h) * C;
    part[u] = 0.0f;
  }
  for (int64_t c0 = 0; c0 < C; c0 += (int64_t)G * 4) {  // C % 4 == 0 on this path
    const int64_t c = c0 + (int64_t)sub * 4;
    const bool act = c < C;
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = act ? *reinterpret_cast<const float4 *>(xr[u] + c) : float4{0.f, 0.f, 0.f, 0.f};
      b[u] = act ? *reinterpret_cast<const float4 *>(gr[u] + c) : float4{0.f, 0.f, 0.f, 0.f};
    }
    // ripple: lane k takes the running sums from lane k-1 and adds its own four products, in channel order
#pragma unroll
    for (int k = 0; k < G; ++k) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float prev = __shfl(part[u], k == 0 ? G - 1 : k - 1, G);  // (k == 0: the sum carried over from c0 - G*4)
        if (sub == k && act) {
          float r = prev;
          r = __fadd_rn(r, __fmul_rn(a[u].x, b[u].x));
          r = __fadd_rn(r, __fmul_rn(a[u].y, b[u].y));
          r = __fadd_rn(r, __fmul_rn(a[u].z, b[u].z));
          r = __fadd_rn(r, __fmul_rn(a[u].w, b[u].w));
          part[u] = r;
        } else if (sub == k) {
          part[u] = prev;  // past the end of the strip: just carry the sum along
        }
      }
    }
  }
  // the last lane of the ripple holds the complete dot
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float dot = __shfl(part[u], G - 1, G);
    if (sub == u && base + u < total) gw[base + u] = dot;
  }
}
#endif

__global__ __launch_bounds__(kBlock) void fill_i64_kernel(int64_t *p, int64_t n, int64_t v) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < n; i += stride) p[i] = v;
}

static inline int64_t grid_for(int64_t n) {
  int64_t g = ceil_div(n, kBlock);
  if (g > 2048) g = 2048;
  return g < 1 ? 1 : g;
}

static inline int group_log2(int64_t K) {
  int l = 0;
  while (l < 6 && ((int64_t)1 << l) < K) ++l;
  return l;
}

template <typename T, bool MEAN>
static int launch_gather(const void *gout, const int64_t *ids, const int64_t *rowptr, int64_t E,
                         int64_t K, void *gin, hipStream_t s) {
  using S = typename TT<T>::S;
  if (E == 0 || K == 0) return GGL_OK;
  const int logL = group_log2(K);
  const int64_t rows_per_block = kWavesPerBlock * (kWave >> logL);
  GGL_LAUNCH((row_gather_kernel<T, 1, MEAN>), ceil_div(E, rows_per_block), kBlock, s,
             static_cast<const S *>(gout), ids, rowptr, E, K, logL, static_cast<S *>(gin));
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_fill_i64(int64_t *p, int64_t n, int64_t v, void *stream) {
  if (n <= 0) return GGL_OK;
  GGL_REQUIRE(p != nullptr, GGL_EINVAL, "p is NULL");
  GGL_LAUNCH((fill_i64_kernel), grid_for(n), kBlock, as_stream(stream), p, n, v);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_segment_sum_bwd(int dtype, const void *gout, const int64_t *ids, int64_t E,
                                   int64_t K, void *gin, void *stream) {
  GGL_REQUIRE(E >= 0 && K >= 0, GGL_EINVAL, "negative size");
  GGL_REQUIRE((gout && ids && gin) || E * K == 0, GGL_EINVAL, "NULL pointer");
  hipStream_t s = as_stream(stream);
  switch (dtype_size(dtype)) {  // a pure row copy: only the element width matters
    case 1: return launch_gather<uint8_t, false>(gout, ids, nullptr, E, K, gin, s);
    case 2: return launch_gather<int16_t, false>(gout, ids, nullptr, E, K, gin, s);
    case 4: return launch_gather<int32_t, false>(gout, ids, nullptr, E, K, gin, s);
    case 8: return launch_gather<int64_t, false>(gout, ids, nullptr, E, K, gin, s);
    default: set_error("unsupported dtype code %d", dtype); return GGL_EDTYPE;
  }
}

extern "C" int ggl_segment_mean_bwd(int dtype, const void *gout, const int64_t *ids,
                                    const int64_t *rowptr, int64_t E, int64_t K, void *gin,
                                    void *stream) {
  GGL_REQUIRE(E >= 0 && K >= 0, GGL_EINVAL, "negative size");
  GGL_REQUIRE((gout && ids && gin && rowptr) || E * K == 0, GGL_EINVAL, "NULL pointer");
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case GGL_F32: return launch_gather<float, true>(gout, ids, rowptr, E, K, gin, s);
    case GGL_F64: return launch_gather<double, true>(gout, ids, rowptr, E, K, gin, s);
    case GGL_F16: return launch_gather<f16_t, true>(gout, ids, rowptr, E, K, gin, s);
    case GGL_BF16: return launch_gather<bf16_t, true>(gout, ids, rowptr, E, K, gin, s);
    default: set_error("mean backward needs a floating dtype (got code %d)", dtype); return GGL_EDTYPE;
  }
}

extern "C" int ggl_segment_max_bwd(int dtype, const void *gout, const int64_t *arg, int64_t E,
                                   int64_t N, int64_t K, void *gin, void *stream) {
  GGL_REQUIRE(E >= 0 && K >= 0 && N >= 0, GGL_EINVAL, "negative size");
  const size_t es = dtype_size(dtype);
  GGL_REQUIRE(es != 0, GGL_EDTYPE, "unsupported dtype code %d", dtype);
  hipStream_t s = as_stream(stream);
  if (E * K > 0) {
    GGL_REQUIRE(gin != nullptr, GGL_EINVAL, "gin is NULL");
    GGL_HIP_CHECK(hipMemsetAsync(gin, 0, (size_t)E * K * es, s));
  }
  const int64_t total = N * K;
  if (total == 0 || E == 0) return GGL_OK;
  GGL_REQUIRE(gout && arg, GGL_EINVAL, "gout/arg is NULL");
  switch (es) {
    case 1: GGL_LAUNCH((max_scatter_kernel<uint8_t>), grid_for(total), kBlock, s, (const uint8_t *)gout, arg, total, K, E, (uint8_t *)gin); break;
    case 2: GGL_LAUNCH((max_scatter_kernel<uint16_t>), grid_for(total), kBlock, s, (const uint16_t *)gout, arg, total, K, E, (uint16_t *)gin); break;
    case 4: GGL_LAUNCH((max_scatter_kernel<uint32_t>), grid_for(total), kBlock, s, (const uint32_t *)gout, arg, total, K, E, (uint32_t *)gin); break;
    default: GGL_LAUNCH((max_scatter_kernel<uint64_t>), grid_for(total), kBlock, s, (const uint64_t *)gout, arg, total, K, E, (uint64_t *)gin); break;
  }
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

// ---- column sums of a row-major [N,K] f32 matrix (bias gradient), two deterministic stages ----------
// stage 1: block b owns rows [b*R, (b+1)*R); its threads are G = 256 / Kp groups of Kp lanes
//          (Kp = min(K, 256)); group j accumulates rows b*R + j, + G, ... for its columns, in order,
//          and writes partial[(b*G + j), k].  stage 2: one thread per column adds the partials in
//          order.  No atomics, no LDS: the result does not depend on scheduling.
__global__ __launch_bounds__(kBlock) void colsum_stage1_kernel(const float *__restrict__ g, int64_t N,
                                                               int64_t K, int64_t nblocks,
                                                               int64_t rows_per_block,
                                                               int kp, int groups,
                                                               float *__restrict__ partial) {
  const int j = threadIdx.x / kp;
  const int k0 = threadIdx.x - j * kp;
  if (j >= groups) return;
  const int64_t r0 = block_id() * rows_per_block;
  if (block_id() >= nblocks) return;  // padding block of a folded grid
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  for (int64_t k = k0; k < K; k += kp) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // 4 independent chains, combined in a fixed order
    int64_t r = r0 + j;
    for (; r + 3 * (int64_t)groups < r1; r += 4 * (int64_t)groups) {
      a0 = __fadd_rn(a0, g[r * K + k]);
      a1 = __fadd_rn(a1, g[(r + groups) * K + k]);
      a2 = __fadd_rn(a2, g[(r + 2 * (int64_t)groups) * K + k]);
      a3 = __fadd_rn(a3, g[(r + 3 * (int64_t)groups) * K + k]);
    }
    for (; r < r1; r += groups) a0 = __fadd_rn(a0, g[r * K + k]);
    partial[(block_id() * groups + j) * K + k] = __fadd_rn(__fadd_rn(a0, a1), __fadd_rn(a2, a3));
  }
}

__global__ __launch_bounds__(kBlock) void colsum_stage2_kernel(const float *__restrict__ partial,
                                                               int64_t P, int64_t K,
                                                               float *__restrict__ out) {
  const int64_t k = thread_id();
  if (k >= K) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // 4 independent chains, combined in a fixed order
  int64_t p = 0;
  for (; p + 4 <= P; p += 4) {
    a0 = __fadd_rn(a0, partial[p * K + k]);
    a1 = __fadd_rn(a1, partial[(p + 1) * K + k]);
    a2 = __fadd_rn(a2, partial[(p + 2) * K + k]);
    a3 = __fadd_rn(a3, partial[(p + 3) * K + k]);
  }
  for (; p < P; ++p) a0 = __fadd_rn(a0, partial[p * K + k]);
  out[k] = __fadd_rn(__fadd_rn(a0, a1), __fadd_rn(a2, a3));
}

constexpr int64_t kColsumSerial = 64;  // partial rows the last stage may walk serially per column

static inline void colsum_geometry(int64_t N, int64_t K, int *kp, int *groups, int64_t *blocks,
                                   int64_t *rows_per_block) {
  *kp = (int)(K < kBlock ? (K > 0 ? K : 1) : kBlock);
  *groups = kBlock / *kp;
  int64_t b = ceil_div(N > 0 ? N : 1, (int64_t)*groups * 64);  // >= 64 rows per group
  if (b > 512) b = 512;  // stage 2 walks blocks * groups partials per column serially
  if (b < 1) b = 1;
  *blocks = b;
  *rows_per_block = ceil_div(N > 0 ? N : 1, b);
}

extern "C" size_t ggl_colsum_workspace_bytes(int64_t N, int64_t K) {
  int kp, groups;
  int64_t blocks, rpb;
  colsum_geometry(N, K, &kp, &groups, &blocks, &rpb);
  const int64_t P = blocks * groups;
  size_t b = (size_t)P * (size_t)(K > 0 ? K : 1) * sizeof(float);
  b = (b + 255) & ~(size_t)255;
  if (P > kColsumSerial && P * 4 <= N) b += ggl_colsum_workspace_bytes(P, K);  // partials reduced the same way
  return b;
}

extern "C" int ggl_colsum_f32(const float *g, int64_t N, int64_t K, float *out, void *workspace,
                              size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(N >= 0 && K >= 0, GGL_EINVAL, "negative size");
  if (K == 0) return GGL_OK;
  GGL_REQUIRE(out != nullptr && (g != nullptr || N == 0), GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_colsum_workspace_bytes(N, K), GGL_EWORKSPACE,
              "colsum workspace too small");
  int kp, groups;
  int64_t blocks, rpb;
  colsum_geometry(N, K, &kp, &groups, &blocks, &rpb);
  hipStream_t s = as_stream(stream);
  float *partial = static_cast<float *>(workspace);
  GGL_LAUNCH((colsum_stage1_kernel), blocks, kBlock, s, g, N, K, blocks, rpb, kp, groups, partial);
  GGL_LAUNCH_CHECK();
  const int64_t P = blocks * groups;
  if (P > kColsumSerial && P * 4 <= N) {  // still many partial rows: reduce them with the same two stages
    size_t off = ((size_t)P * (size_t)K * sizeof(float) + 255) & ~(size_t)255;
    return ggl_colsum_f32(partial, P, K, out, static_cast<char *>(workspace) + off, workspace_bytes - off,
                          stream);
  }
  GGL_LAUNCH((colsum_stage2_kernel), ceil_div(K, kBlock), kBlock, s, (const float *)partial, P, K, out);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_bspmm_grad_w(const int64_t *index, const float *x, const float *g, int64_t E,
                                int64_t H, int64_t C, float *gw, void *stream) {
  GGL_REQUIRE(E >= 0 && H > 0 && C > 0, GGL_EINVAL, "bad sizes");
  if (E == 0) return GGL_OK;
  GGL_REQUIRE(index && x && g && gw, GGL_EINVAL, "NULL pointer");
#ifndef GGL_EMULATE
  // (measured at products size: 8 x 44 channels 165 -> 127 ms forward+backward with the ripple; beyond 16 lanes the
  // lane-to-lane ripple costs more than the strided loads it saves — 1 x 256: 103 -> 171 ms — so wider heads
  // stay on the thread-per-item kernel; a butterfly would be faster still, 57 ms, but not bit-exact)
  if (C > 16 && C <= 64 && C % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15u) == 0 && !options().force_generic) {
    int logg = 2;
    while (logg < 4 && ((int64_t)4 << logg) < C) ++logg;  // lanes per item: next power of two >= C / 4, 4..16
    const int64_t groups = ceil_div(E * H, (int64_t)kGradWPerGroup);
    const int64_t wgrid = ceil_div(groups << logg, (int64_t)kBlock);
    hipStream_t s = as_stream(stream);
    if (logg <= 2) GGL_LAUNCH((bspmm_grad_w_wide_kernel<2>), wgrid, kBlock, s, index, x, g, E, H, C, gw);
    else if (logg == 3) GGL_LAUNCH((bspmm_grad_w_wide_kernel<3>), wgrid, kBlock, s, index, x, g, E, H, C, gw);
    else GGL_LAUNCH((bspmm_grad_w_wide_kernel<4>), wgrid, kBlock, s, index, x, g, E, H, C, gw);
    GGL_LAUNCH_CHECK();
    return GGL_OK;
  }
#endif
  GGL_LAUNCH((bspmm_grad_w_kernel), grid_for(E * H), kBlock, as_stream(stream), index, x, g, E, H, C,
             gw);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}
