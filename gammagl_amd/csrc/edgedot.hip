// gammagl_amd/csrc/edgedot.hip — bspmm's weight gradient on the destination-sorted plan:
//   ggl_bspmm_grad_w_sorted : gw[e,h] = sum_c x[src_e,h,c] * g[dst_e,h,c]   (cpu/bspmm_sum_cpu.cpp:95-107)
//
// The reference sums over c serially (rounded multiply, rounded add, c ascending) and the golden weight gradients pin
// that order bit for bit, so the dot of an (edge, head) item is one dependent chain: it has to live in ONE lane.  The
// thread-per-item kernel of backward.hip does exactly that and reads its two strips 16 bytes at a time — a wavefront
// then touches 64 different rows per load, C/4 times over, which thrashes the 32 KiB L1 of a CU for C > 16 (every
// 128-byte line is fetched up to 8 times from L2): products-sized graph, H = 1, C = 256: 69 ms where the forward
// SpMM — the same gather volume — takes 16.4 ms.
//
// Here the loads are coalesced and the transposition happens in LDS: a workgroup owns 256 consecutive items of the
// DESTINATION-SORTED order (so the g strips of a batch are a handful of rows: each lane reads its own straight from
// memory and L1 serves the lanes that share a row; only x[src] is a random gather, exactly as in the forward walk),
// stages 32-column slabs of the items' x strips into an LDS tile with 16-byte loads in which 8 consecutive lanes cover
// 128 contiguous bytes — the next slab's loads in flight while the current one is folded — and every lane folds ITS
// item's 32 products in order from LDS (row stride 36 floats: the 16-byte reads of a 16-lane pass fall into distinct
// banks).
// Results go to gw[perm[p], h], the caller's edge order.  A launch may cover a column range [c_lo, c_hi) only,
// taking the chain so far from `carry_in` and leaving it in `carry_out` (both in SORTED order: coalesced): the host
// runs wide heads as launches over 64-column blocks like the forward (launch_f32_cols, reduce.hip) — same serial
// order, and the 256-byte slices keep 4x as many hub rows in L2; only the last block scatters through perm.
#include "common.hpp"

namespace ggl {

// float4s per strip per slab: 8 (32 columns; tile row stride 36 floats: the 16-byte reads of a 16-lane pass fall into
// distinct banks).  (12-piece slabs — a 44-channel head as ONE slab — were tried: 180 registers, two workgroups per CU,
// 8 x 44 forward+backward 82 -> 88 ms.)
template <int Q> struct DotTile { static constexpr int ld = Q * 4 + 4; };

#ifndef GGL_EMULATE
// One slab = NQ float4s (4 NQ columns) of every item's x strip.  NQ is a compile-time constant: the slab lives in
// registers between its loads and its LDS stores, nothing is indexed at run time (the first version indexed a shared
// register array under a run-time bound and the backend put it in scratch memory: 2x slower than the kernel it replaced).
template <int NQ>
__device__ __forceinline__ void slab_load(int tid, const int64_t *sx, const float *__restrict__ x, int64_t c0,
                                          float4 (&v)[NQ]) {
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int f = tid + kBlock * j, item = f / NQ, part = f - item * NQ;   // 8 consecutive lanes = 128 contiguous bytes
    const int64_t ox = sx[item];
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ox >= 0) v[j] = *reinterpret_cast<const float4 *>(x + ox + c0 + part * 4);
  }
}
// the same on the first NQ entries of the pipeline's 8-entry register slab (the narrower last slab of a strip)
template <int NQ, int Q>
__device__ __forceinline__ void slab_load_into(int tid, const int64_t *sx, const float *__restrict__ x, int64_t c0,
                                               float4 (&v)[Q]) {
  static_assert(NQ <= Q, "tail slab wider than the pipeline's");
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int f = tid + kBlock * j, item = f / NQ, part = f - item * NQ;
    const int64_t ox = sx[item];
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ox >= 0) v[j] = *reinterpret_cast<const float4 *>(x + ox + c0 + part * 4);
  }
}
template <int NQ, int LD, int Q>
__device__ __forceinline__ void slab_store_from(int tid, const float4 (&v)[Q], float (*tx)[LD]) {
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int f = tid + kBlock * j, item = f / NQ, part = f - item * NQ;
    *reinterpret_cast<float4 *>(&tx[item][part * 4]) = v[j];
  }
}
template <int NQ, int LD>
__device__ __forceinline__ void slab_store(int tid, const float4 (&v)[NQ], float (*tx)[LD]) {
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int f = tid + kBlock * j, item = f / NQ, part = f - item * NQ;
    *reinterpret_cast<float4 *>(&tx[item][part * 4]) = v[j];
  }
}
// this lane's own item: its x slab from LDS, its g slab straight from memory (the g strips of a batch of sorted
// positions are a handful of destination rows: lanes that share a row read the same addresses, L1 serves them)
template <int NQ, int LD>
__device__ __forceinline__ float slab_fold(int tid, float (*tx)[LD], const float *__restrict__ gp, bool valid,
                                           float acc) {
  float4 gg[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    gg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) gg[k] = *reinterpret_cast<const float4 *>(gp + k * 4);
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const float4 a = *reinterpret_cast<const float4 *>(&tx[tid][k * 4]);
    acc = __fadd_rn(acc, __fmul_rn(a.x, gg[k].x));
    acc = __fadd_rn(acc, __fmul_rn(a.y, gg[k].y));
    acc = __fadd_rn(acc, __fmul_rn(a.z, gg[k].z));
    acc = __fadd_rn(acc, __fmul_rn(a.w, gg[k].w));
  }
  return acc;
}
// TAIL = float4s of the strip's last, narrower slab (0 = the column range is whole slabs): a template parameter, so each
// instantiation carries one tail shape (a run-time switch over seven shapes inside one kernel cost 188 registers
// instead of ~120, i.e. half the resident wavefronts)
template <int Q, int TAIL>
__global__ __launch_bounds__(kBlock) void bspmm_grad_w_sorted_kernel(
    const int32_t *__restrict__ col, const int32_t *__restrict__ rowidx, const int32_t *__restrict__ perm,
    const float *__restrict__ x, const float *__restrict__ g, int64_t total, int64_t H, int64_t C, int64_t c_lo,
    int64_t c_hi, const float *__restrict__ carry_in, float *__restrict__ carry_out, float *__restrict__ gw) {
  constexpr int kDotQ = Q, kDotLd = DotTile<Q>::ld;
  __shared__ __attribute__((aligned(16))) float tx[kBlock][kDotLd];   // Q = 8: 36 KiB, four workgroups per CU
  __shared__ int64_t sx[kBlock];
  const int tid = threadIdx.x;
  const int64_t base = block_id() * (int64_t)kBlock;
  if (base >= total) return;
  const int64_t i = base + tid;
  const bool valid = i < total;
  int64_t p = 0, h = 0;
  if (valid) {
    p = i / H;
    h = i - p * H;
  }
  sx[tid] = valid ? ((int64_t)col[p] * H + h) * C : (int64_t)-1;
  const float *gp = g + (valid ? ((int64_t)rowidx[p] * H + h) * C : 0);
  float acc = (valid && carry_in) ? carry_in[i] : 0.0f;   // the chain so far (sorted order: coalesced)
  __syncthreads();
  int64_t c0 = c_lo;
  const int64_t n_full = (c_hi - c_lo) / (kDotQ * 4);
  // software pipeline: the next slab's loads (a full one, or the tail) are in flight while the current one is folded
  float4 cur[kDotQ];
  if (n_full > 0) slab_load<kDotQ>(tid, sx, x, c0, cur);
  else if constexpr (TAIL > 0) slab_load_into<TAIL>(tid, sx, x, c0, cur);
  for (int64_t s = 0; s < n_full; ++s) {
    slab_store<kDotQ, kDotLd>(tid, cur, tx);
    __syncthreads();
    if (s + 1 < n_full) slab_load<kDotQ>(tid, sx, x, c0 + kDotQ * 4, cur);
    else if constexpr (TAIL > 0) slab_load_into<TAIL>(tid, sx, x, c0 + kDotQ * 4, cur);
    acc = slab_fold<kDotQ, kDotLd>(tid, tx, gp + c0, valid, acc);
    __syncthreads();
    c0 += kDotQ * 4;
  }
  if constexpr (TAIL > 0) {
    slab_store_from<TAIL, kDotLd>(tid, cur, tx);
    __syncthreads();
    acc = slab_fold<TAIL, kDotLd>(tid, tx, gp + c0, valid, acc);
  }
  if (!valid) return;
  if (carry_out) carry_out[i] = acc;
  else gw[(perm ? (int64_t)perm[p] : p) * H + h] = acc;
}
#endif

// the same walk one item per thread, strips read in place: the host-emulation build's stand-in (its "threads" run one
// after another: no LDS, no barriers) and the route for strips that are not made of aligned float4s
__global__ __launch_bounds__(kBlock) void bspmm_grad_w_sorted_plain_kernel(
    const int32_t *__restrict__ col, const int32_t *__restrict__ rowidx, const int32_t *__restrict__ perm,
    const float *__restrict__ x, const float *__restrict__ g, int64_t total, int64_t H, int64_t C,
    float *__restrict__ gw) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < total; i += stride) {
    const int64_t p = i / H, h = i - p * H;
    const float *xr = x + ((int64_t)col[p] * H + h) * C;
    const float *gr = g + ((int64_t)rowidx[p] * H + h) * C;
    const int64_t oi = (perm ? (int64_t)perm[p] : p) * H + h;
    float acc = 0.0f;
    for (int64_t c = 0; c < C; ++c) acc = __fadd_rn(acc, __fmul_rn(xr[c], gr[c]));
    gw[oi] = acc;
  }
}

}  // namespace ggl

using namespace ggl;

// gw[e, h] for the edges of `plan` (destination-sorted; plan->perm maps sorted positions back to the caller's edge
// order, NULL = already sorted), `col` / `rowidx` = source / destination node of every sorted position.
static int64_t dot_block_width(int64_t E, int64_t N, int64_t C) {   // C = one launch
#ifdef GGL_EMULATE
  (void)E; (void)N;
  return C;
#else
  // the block width is an A/B knob (ggl_set_option "col_block"): the slab loads are float4 and the tail template covers
  // (width % 32) / 4 quads, so a width that is not a multiple of 4 would drop columns from the dot — such a setting
  // runs as ONE launch here instead
  const int64_t bw = options().col_block;
  if (bw > 0 && bw % 4 == 0 && C % 4 == 0 && C >= 2 * bw && N > 0 &&
      E >= options().col_block_min_degree * N && E >= options().col_block_min_edges)
    return bw;
  return C;
#endif
}

// bytes of `scratch` the call below wants (0: it runs as one launch and needs none)
extern "C" size_t ggl_bspmm_grad_w_sorted_scratch_bytes(int64_t E, int64_t N, int64_t H, int64_t C) {
  return dot_block_width(E, N, C) < C ? (size_t)E * (size_t)H * sizeof(float) : 0;
}

extern "C" int ggl_bspmm_grad_w_sorted(const ggl_segplan_t *plan, const int32_t *col, const int32_t *rowidx,
                                       const float *x, const float *g, int64_t H, int64_t C, float *gw,
                                       float *scratch, void *stream) {
  GGL_REQUIRE(plan != nullptr && H > 0 && C > 0 && plan->E >= 0, GGL_EINVAL, "bad sizes");
  const int64_t E = plan->E;
  if (E == 0) return GGL_OK;
  GGL_REQUIRE(col && rowidx && x && g && gw, GGL_EINVAL, "NULL pointer");
  const int64_t total = E * H;
  hipStream_t s = as_stream(stream);
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15u) == 0 &&
                   !options().force_generic;
#ifndef GGL_EMULATE
  if (vec) {
    // wide strips in 64-column blocks (see the header): only where there are hub rows to keep in L2
    const int64_t bw = scratch ? dot_block_width(E, plan->N, C) : C;
    for (int64_t c0 = 0; c0 < C; c0 += bw) {
      const int64_t c1 = (c0 + bw < C) ? c0 + bw : C;
      const float *cin = c0 > 0 ? scratch : nullptr;
      float *cout = c1 < C ? scratch : nullptr;
#define GGL_DOT_LAUNCH(T)                                                                                         \
  GGL_LAUNCH((bspmm_grad_w_sorted_kernel<8, T>), ceil_div(total, (int64_t)kBlock), kBlock, s, col, rowidx, plan->perm, x, \
             g, total, H, C, c0, c1, cin, cout, gw)
      switch ((int)(((c1 - c0) % 32) / 4)) {
        case 0: GGL_DOT_LAUNCH(0); break;
        case 1: GGL_DOT_LAUNCH(1); break;
        case 2: GGL_DOT_LAUNCH(2); break;
        case 3: GGL_DOT_LAUNCH(3); break;
        case 4: GGL_DOT_LAUNCH(4); break;
        case 5: GGL_DOT_LAUNCH(5); break;
        case 6: GGL_DOT_LAUNCH(6); break;
        default: GGL_DOT_LAUNCH(7); break;
      }
#undef GGL_DOT_LAUNCH
      GGL_LAUNCH_CHECK();
    }
    return GGL_OK;
  }
#endif
  (void)vec;
  int64_t grid = ceil_div(total, (int64_t)kBlock);
  if (grid > 4096) grid = 4096;
  (void)scratch;
  GGL_LAUNCH((bspmm_grad_w_sorted_plain_kernel), grid, kBlock, s, col, rowidx, plan->perm, x, g, total, H, C, gw);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}
