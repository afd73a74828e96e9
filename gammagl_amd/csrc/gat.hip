// gammagl_amd/csrc/gat.hip — fused GAT edge-softmax + weighted aggregate, one kernel per direction.
//
// Replaces (a) the external dgNN GATConvFuse CUDA kernel that FusedGATConv calls
// (layers/conv/fusedgat_conv.py:70-71,121 — not in the reference tree, parity unpinned) and (b) the
// unfused chain GATConv.forward runs today (layers/conv/gat_conv.py:103-112 + utils/softmax.py:29-35):
// 2 gathers [E,H,C] + concat + reduce -> leaky_relu -> segment_max -> gather -> exp -> segment_sum ->
// gather -> divide -> gather [E,H,C] * alpha -> segment_sum, i.e. three segment passes and five
// [E,.] intermediates in HBM.  Here a lane group owns one destination row, each lane VEC channels of
// one head; it walks the row three times (max, denominator, weighted sum).  The first two walks only
// touch el[col[p],h] (N*H floats: cache resident), the third streams the feature rows once.  The
// arithmetic follows the in-tree math exactly: max with strict <, denominator summed in edge order,
// alpha = exp(s - m) / (d + 1e-16), message = x * alpha (rounded), sum in edge order.
// No LDS, no shuffles, no atomics: lanes of the same head recompute the (cheap) scalar softmax terms
// redundantly instead of exchanging them.
// Roofline: HBM; algorithmic bytes per edge = 4*H*C (feature row) + 4 (col) + 4*H (el row).
#include "common.hpp"

#ifndef GGL_EMULATE
#define GGL_EXPF(x) expf(x)
#else
#define GGL_EXPF(x) std::exp(x)
#endif

namespace ggl {

struct GatArgs {
  const int64_t *rowptr;
  const int32_t *col;
  const float *el, *er, *x, *g, *out;
  float slope;
  int64_t N, H, C, K;
  int logL, swizzle;
  int64_t nblocks;
  float *y, *rowmax, *rowden;
  float *alpha, *de, *ger;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : __fmul_rn(v, slope); }

template <int VEC>
__global__ __launch_bounds__(kBlock) void gat_fwd_kernel(const GatArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t blk = xcd_remap((int64_t)blockIdx.x, a.nblocks, a.swizzle);
  const int L = 1 << a.logL;
  const int64_t row = (blk * kWavesPerBlock + wave) * (kWave >> a.logL) + (lane >> a.logL);
  if (row >= a.N) return;
  const int li = lane & (L - 1);
  const int64_t beg = a.rowptr[row], end = a.rowptr[row + 1];
  const int64_t H = a.H, K = a.K;
  for (int64_t kk = (int64_t)li * VEC; kk < K; kk += (int64_t)L * VEC) {
    const int64_t h = kk / a.C;
    const float er_i = a.er[row * H + h];
    // walk 1: m = max_p s   (unsorted_segment_max: lowest() fill, strict <)
    float m = -FLT_MAX;
    for (int64_t p = beg; p < end; ++p) {
      const float s = lrelu(__fadd_rn(a.el[(int64_t)a.col[p] * H + h], er_i), a.slope);
      if (m < s) m = s;
    }
    // walk 2: d = sum_p exp(s - m) in edge order (unsorted_segment_sum)
    float d = 0.0f;
    for (int64_t p = beg; p < end; ++p) {
      const float s = lrelu(__fadd_rn(a.el[(int64_t)a.col[p] * H + h], er_i), a.slope);
      d = __fadd_rn(d, GGL_EXPF(__fadd_rn(s, -m)));
    }
    const float den = __fadd_rn(d, 1e-16f);
    // walk 3: out = sum_p (exp(s - m) / den) * x[col[p]]
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    int64_t p = beg;
    for (; p + 2 <= end; p += 2) {  // two feature rows in flight
      const int64_t c0 = a.col[p], c1 = a.col[p + 1];
      float v0[VEC], v1[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { v0[i] = a.x[c0 * K + kk + i]; v1[i] = a.x[c1 * K + kk + i]; }
      const float s0 = lrelu(__fadd_rn(a.el[c0 * H + h], er_i), a.slope);
      const float s1 = lrelu(__fadd_rn(a.el[c1 * H + h], er_i), a.slope);
      const float a0 = __fdiv_rn(GGL_EXPF(__fadd_rn(s0, -m)), den);
      const float a1 = __fdiv_rn(GGL_EXPF(__fadd_rn(s1, -m)), den);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = __fadd_rn(acc[i], __fmul_rn(v0[i], a0));
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = __fadd_rn(acc[i], __fmul_rn(v1[i], a1));
    }
    for (; p < end; ++p) {
      const int64_t c0 = a.col[p];
      const float s0 = lrelu(__fadd_rn(a.el[c0 * H + h], er_i), a.slope);
      const float a0 = __fdiv_rn(GGL_EXPF(__fadd_rn(s0, -m)), den);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = __fadd_rn(acc[i], __fmul_rn(a.x[c0 * K + kk + i], a0));
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) a.y[row * K + kk + i] = acc[i];
    if (kk == h * a.C) {  // first lane of the head records the softmax statistics
      a.rowmax[row * H + h] = m;
      a.rowden[row * H + h] = d;
    }
  }
}

// Backward, destination-major: one lane per (row, head).  With alpha_p = softmax_p(s_p):
//   dalpha_p = <g[i,h,:], x[col[p],h,:]>,   sum_p alpha_p dalpha_p = <g[i,h,:], out[i,h,:]>
//   ds_p = alpha_p (dalpha_p - <g,out>),    de_p = ds_p * LeakyReLU'(el+er)
//   ger[i,h] = sum_p de_p;   alpha / de are written in forward sorted positions for the source pass.
__global__ __launch_bounds__(kBlock) void gat_bwd_dst_kernel(const GatArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int L = 1 << a.logL;  // lanes per row = pow2 >= H
  const int64_t row = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * (kWave >> a.logL) + (lane >> a.logL);
  if (row >= a.N) return;
  const int64_t H = a.H, C = a.C, K = a.K;
  for (int64_t h = lane & (L - 1); h < H; h += L) {
    const int64_t beg = a.rowptr[row], end = a.rowptr[row + 1];
    const float er_i = a.er[row * H + h];
    const float m = a.rowmax[row * H + h];
    const float den = __fadd_rn(a.rowden[row * H + h], 1e-16f);
    const float *gi = a.g + row * K + h * C;
    const float *oi = a.out + row * K + h * C;
    float dot = 0.0f;
    for (int64_t c = 0; c < C; ++c) dot = __fadd_rn(dot, __fmul_rn(gi[c], oi[c]));
    float gacc = 0.0f;
    for (int64_t p = beg; p < end; ++p) {
      const int64_t src = a.col[p];
      const float raw = __fadd_rn(a.el[src * H + h], er_i);
      const float s = lrelu(raw, a.slope);
      const float al = __fdiv_rn(GGL_EXPF(__fadd_rn(s, -m)), den);
      const float *xj = a.x + src * K + h * C;
      float da = 0.0f;
      for (int64_t c = 0; c < C; ++c) da = __fadd_rn(da, __fmul_rn(gi[c], xj[c]));
      const float ds = __fmul_rn(al, __fadd_rn(da, -dot));
      const float dv = raw > 0.0f ? ds : __fmul_rn(ds, a.slope);
      a.alpha[p * H + h] = al;
      a.de[p * H + h] = dv;
      gacc = __fadd_rn(gacc, dv);
    }
    a.ger[row * H + h] = gacc;
  }
}

static inline int pow2_log2(int64_t v) {
  int l = 0;
  while (l < 6 && ((int64_t)1 << l) < v) ++l;
  return l;
}

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_gat_fused_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el,
                                 const float *er, const float *x, float slope, int64_t H, int64_t C,
                                 float *out, float *rowmax, float *rowden, void *stream) {
  GGL_REQUIRE(plan && plan->rowptr, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(H > 0 && C > 0, GGL_EINVAL, "H and C must be positive");
  const int64_t N = plan->N;
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(er && out && rowmax && rowden, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((col && el && x) || plan->E == 0, GGL_EINVAL, "NULL pointer");
  GatArgs a{};
  a.rowptr = plan->rowptr; a.col = col; a.el = el; a.er = er; a.x = x; a.slope = slope;
  a.N = N; a.H = H; a.C = C; a.K = H * C; a.y = out; a.rowmax = rowmax; a.rowden = rowden;
  a.swizzle = (int)options().xcd_swizzle;
  const bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) && !options().force_generic;
  const int vec = vec4 ? 4 : 1;
  a.logL = pow2_log2(ceil_div(a.K, vec));
  a.nblocks = ceil_div(N, (int64_t)kWavesPerBlock * (kWave >> a.logL));
  GGL_REQUIRE(a.nblocks < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  if (vec4) GGL_LAUNCH((gat_fwd_kernel<4>), a.nblocks, kBlock, as_stream(stream), a);
  else GGL_LAUNCH((gat_fwd_kernel<1>), a.nblocks, kBlock, as_stream(stream), a);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_gat_fused_bwd_dst(const ggl_segplan_t *plan, const int32_t *col, const float *el,
                                     const float *er, const float *x, const float *g,
                                     const float *out, const float *rowmax, const float *rowden,
                                     float slope, int64_t H, int64_t C, float *alpha, float *de,
                                     float *ger, void *stream) {
  GGL_REQUIRE(plan && plan->rowptr, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(H > 0 && C > 0, GGL_EINVAL, "H and C must be positive");
  const int64_t N = plan->N;
  if (N == 0) return GGL_OK;
  GGL_REQUIRE(er && g && out && rowmax && rowden && ger, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE((col && el && x && alpha && de) || plan->E == 0, GGL_EINVAL, "NULL pointer");
  GatArgs a{};
  a.rowptr = plan->rowptr; a.col = col; a.el = el; a.er = er; a.x = x; a.g = g; a.out = out;
  a.slope = slope; a.N = N; a.H = H; a.C = C; a.K = H * C;
  a.rowmax = const_cast<float *>(rowmax); a.rowden = const_cast<float *>(rowden);
  a.alpha = alpha; a.de = de; a.ger = ger;
  a.logL = pow2_log2(H);
  a.nblocks = ceil_div(N, (int64_t)kWavesPerBlock * (kWave >> a.logL));
  GGL_REQUIRE(a.nblocks < ((int64_t)1 << 31), GGL_EINVAL, "too many rows for one launch");
  GGL_LAUNCH((gat_bwd_dst_kernel), a.nblocks, kBlock, as_stream(stream), a);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

// Source-major half of the backward: two row reductions on the transposed plan, reading alpha / de
// through posT (transposed position -> forward position):
//   gx[j,h,:] = sum_p alpha[posT[p],h] * g[colT[p],h,:]     == ggl_bspmm_sum with perm = posT
//   gel[j,h]  = sum_p de[posT[p],h]                         == ggl_segment_sum with perm = posT
extern "C" int ggl_gat_fused_bwd_src(const ggl_segplan_t *planT, const int32_t *colT,
                                     const int32_t *posT, const float *alpha, const float *de,
                                     const float *g, int64_t H, int64_t C, float *gx, float *gel,
                                     void *stream) {
  GGL_REQUIRE(planT && planT->rowptr, GGL_EINVAL, "planT is NULL");
  ggl_segplan_t p = *planT;
  p.perm = posT;
  int rc = ggl_bspmm_sum(&p, colT, alpha, /*w_by_pos=*/0, g, H, C, gx, stream);
  if (rc) return rc;
  return ggl_segment_sum(GGL_F32, de, &p, H, gel, stream);
}
