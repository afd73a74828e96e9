// gammagl_amd/csrc/epilogue.hip — the elementwise step right after every aggregate:
//   GCNConv: out += bias (gcn_conv.py:105-106); GCNModel: relu -> dropout (models/gcn.py:55-59);
//   SAGEConv: + bias, activation (sage_conv.py:102-106).
// torch runs this as add, clamp, dropout (+ a bool mask) forward and masked_scale, threshold, sum(0)
// backward: six passes over [N,K] per hidden layer (measured 4.0 ms of a 110 ms products-sized step,
// profiles/r1_bench_products_summary.txt).  Here it is one pass each way:
//   forward : y = keep(i) * relu(a + bias) / (1 - p)        keep(i) from Philox4x32-10(seed, offset; i)
//   backward: ga = keep(i) * [relu ? y > 0 : 1] * g / (1 - p): the dropout mask is REDRAWN from the
//             {seed, offset} the forward read (no mask tensor is stored; y only tells where the ReLU
//             clipped), and the bias gradient = column sums of ga accumulated in the same pass (two
//             deterministic stages, as in ggl_colsum_f32).
// The RNG state (seed, offset) lives in device memory and is advanced by a one-thread kernel after every
// forward, so a captured hipGraph draws a fresh mask on each replay.  No LDS, no atomics.
#include "common.hpp"

namespace ggl {

template <int VEC> __device__ __forceinline__ void ldv(const float *__restrict__ p, float (&t)[VEC]) {
  if (VEC == 4) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    t[0] = v.x; t[1 % VEC] = v.y; t[2 % VEC] = v.z; t[3 % VEC] = v.w;
  } else {
    t[0] = p[0];
  }
}
template <int VEC> __device__ __forceinline__ void stv(float *__restrict__ p, const float (&t)[VEC]) {
  if (VEC == 4) {
    float4 o;
    o.x = t[0]; o.y = t[1 % VEC]; o.z = t[2 % VEC]; o.w = t[3 % VEC];
    *reinterpret_cast<float4 *>(p) = o;
  } else {
    p[0] = t[0];
  }
}
constexpr int kRowUnroll = 4;  // rows in flight per lane: the loads of 4 rows are issued before any is used

// Threads of a block are `groups` groups of `kp` lanes; a lane owns VEC consecutive columns (loaded once
// from bias), a group walks rows r0 + j, + groups, ...  Random word for vector (r, c): Philox(r * KV + c).
template <int VEC>
__global__ __launch_bounds__(kBlock) void bias_act_fwd_kernel(const float *__restrict__ a,
                                                              const float *__restrict__ bias,
                                                              const int64_t *__restrict__ rng,
                                                              float *__restrict__ y, int64_t N, int64_t K,
                                                              int64_t nblocks, int64_t rows_per_block, int kp,
                                                              int groups,
                                                              int relu, uint32_t drop_thresh, float scale) {
  const int j = threadIdx.x / kp;
  const int c0 = threadIdx.x - j * kp;
  if (j >= groups) return;
  const int64_t KV = (K + VEC - 1) / VEC;  // vectors per row (VEC = 4 only when K % 4 == 0)
  const int ev = (K % 4 == 0) ? 4 : 1;     // elements per dropout word vector
  const int64_t KVm = K / ev;
  const int64_t r0 = block_id() * rows_per_block;
  if (block_id() >= nblocks) return;  // padding block of a folded grid
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  const uint64_t seed = drop_thresh ? (uint64_t)rng[0] : 0, offset = drop_thresh ? (uint64_t)rng[1] : 0;
  for (int64_t c = c0; c < KV; c += kp) {
    float b[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) b[i] = bias ? bias[c * VEC + i] : 0.0f;
    auto finish = [&](int64_t r, float (&t)[VEC]) {
      uint32_t rw[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (drop_thresh) {
        // the word of element (r, k) depends on K alone (vectors of 4 when K % 4 == 0), not on which kernel
        // variant the pointer alignment selected: the fused SpMM epilogue and the backward draw the same mask
        const U4 u = philox4x32_10((uint64_t)(r * KVm + (c * VEC) / ev), offset, seed);
        rw[0] = u.x; rw[1] = u.y; rw[2] = u.z; rw[3] = u.w;
        if (VEC == 1) rw[0] = rw[(c * VEC) % ev];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = bias ? __fadd_rn(t[i], b[i]) : t[i];
        if (relu) v = (v < 0.0f) ? 0.0f : v;  // NaN stays NaN, as torch.relu / clamp_min(0)
        if (drop_thresh) v = (rw[i] >= drop_thresh) ? __fmul_rn(v, scale) : 0.0f;  // keep with prob 1 - p
        t[i] = v;
      }
      stv<VEC>(y + r * K + c * VEC, t);
    };
    int64_t r = r0 + j;
    for (; r + (int64_t)(kRowUnroll - 1) * groups < r1; r += (int64_t)kRowUnroll * groups) {
      float t[kRowUnroll][VEC];
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) ldv<VEC>(a + (r + (int64_t)u * groups) * K + c * VEC, t[u]);
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) finish(r + (int64_t)u * groups, t[u]);
    }
    for (; r < r1; r += groups) {
      float t[VEC];
      ldv<VEC>(a + r * K + c * VEC, t);
      finish(r, t);
    }
  }
}

__global__ void rng_advance_kernel(int64_t *rng, int64_t inc) {
  if (block_id() == 0 && threadIdx.x == 0) rng[1] += inc;
}

// backward + first stage of the bias gradient.  Same thread geometry as the forward; every lane keeps a
// running column sum of the ga values it produces and writes it to partial[(block * groups + j), :];
// ggl_colsum_f32 (backward.hip) then reduces the [P, K] partial matrix.
template <int VEC, int MASKED>
__global__ __launch_bounds__(kBlock) void bias_act_bwd_kernel(const float *__restrict__ g,
                                                              const float *__restrict__ y,
                                                              float *__restrict__ ga, int64_t N,
                                                              int64_t K, int64_t nblocks,
                                                              int64_t rows_per_block, int kp,
                                                              int groups, float scale,
                                                              const int64_t *__restrict__ rng,
                                                              uint32_t drop_thresh,
                                                              float *__restrict__ partial) {
  const int j = threadIdx.x / kp;
  const int c0 = threadIdx.x - j * kp;
  if (j >= groups) return;
  const int64_t KV = (K + VEC - 1) / VEC;
  const int ev = (K % 4 == 0) ? 4 : 1;
  const int64_t KVm = K / ev;
  const int64_t r0 = block_id() * rows_per_block;
  if (block_id() >= nblocks) return;  // padding block of a folded grid
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  for (int64_t c = c0; c < KV; c += kp) {
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    auto finish = [&](int64_t r, float (&gv)[VEC], const float (&yv)[VEC]) {
      constexpr int masked = MASKED;  // compile-time: the common ReLU case carries no Philox code at all
      uint32_t rw[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (MASKED == 3) {  // the mask the forward drew: same word for vector (r, c)
        const U4 u = philox4x32_10((uint64_t)(r * KVm + (c * VEC) / ev), (uint64_t)rng[1], (uint64_t)rng[0]);
        rw[0] = u.x; rw[1] = u.y; rw[2] = u.z; rw[3] = u.w;
        if (VEC == 1) rw[0] = rw[(c * VEC) % ev];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        // masked: 1 = y > 0 (ReLU, with or without dropout: exact), 2 = y != 0 (dropout without ReLU and
        // without an rng state), 3 = redrawn dropout mask (dropout without ReLU)
        if (masked == 1) gv[i] = (yv[i] > 0.0f) ? __fmul_rn(gv[i], scale) : 0.0f;
        else if (masked == 2) gv[i] = (yv[i] != 0.0f) ? __fmul_rn(gv[i], scale) : 0.0f;
        else if (masked == 3) gv[i] = (rw[i] >= drop_thresh) ? __fmul_rn(gv[i], scale) : 0.0f;
        acc[i] = __fadd_rn(acc[i], gv[i]);  // rows in ascending order, unrolled or not
      }
      stv<VEC>(ga + r * K + c * VEC, gv);
    };
    int64_t r = r0 + j;
    for (; r + (int64_t)(kRowUnroll - 1) * groups < r1; r += (int64_t)kRowUnroll * groups) {
      float gv[kRowUnroll][VEC], yv[kRowUnroll][VEC];
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        ldv<VEC>(g + (r + (int64_t)u * groups) * K + c * VEC, gv[u]);
        if (MASKED == 1 || MASKED == 2) ldv<VEC>(y + (r + (int64_t)u * groups) * K + c * VEC, yv[u]);
        else yv[u][0] = 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) finish(r + (int64_t)u * groups, gv[u], yv[u]);
    }
    for (; r < r1; r += groups) {
      float gv[VEC], yv[VEC];
      ldv<VEC>(g + r * K + c * VEC, gv);
      if (MASKED == 1 || MASKED == 2) ldv<VEC>(y + r * K + c * VEC, yv);
      else yv[0] = 0.0f;
      finish(r, gv, yv);
    }
    if (partial) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) partial[(block_id() * groups + j) * K + c * VEC + i] = acc[i];
    }
  }
}

// launch geometry shared by both directions: kp lanes x groups rows per block step, `blocks` row ranges
static inline void geometry(int64_t N, int64_t K, bool vec4, int *kp, int *groups, int64_t *blocks,
                            int64_t *rpb) {
  const int64_t KV = vec4 ? K / 4 : K;
  *kp = (int)(KV < kBlock ? (KV > 0 ? KV : 1) : kBlock);
  *groups = kBlock / *kp;
  int64_t b = ceil_div(N > 0 ? N : 1, (int64_t)*groups * 8);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  *blocks = b;
  *rpb = ceil_div(N > 0 ? N : 1, b);
}

int rng_advance(int64_t *rng_state, void *stream) {
  GGL_LAUNCH((rng_advance_kernel), 1, 64, as_stream(stream), rng_state, (int64_t)1);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_bias_act_fwd(const float *a, const float *bias, int64_t N, int64_t K, int relu,
                                float p_drop, int64_t *rng_state, float *y, void *stream) {
  GGL_REQUIRE(N >= 0 && K >= 0 && p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "bad arguments");
  const int64_t total = N * K;
  if (total == 0) return GGL_OK;
  GGL_REQUIRE(a && y, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(p_drop == 0.0f || rng_state, GGL_EINVAL, "dropout needs an rng_state");
  // keep when r >= thresh, r uniform on [0, 2^32): P(drop) = thresh / 2^32
  const uint32_t thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  const float scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  hipStream_t s = as_stream(stream);
  const bool vec4 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(y) & 15u) == 0);
  int kp, groups;
  int64_t grid, rpb;
  geometry(N, K, vec4, &kp, &groups, &grid, &rpb);
  if (vec4)
    GGL_LAUNCH((bias_act_fwd_kernel<4>), grid, kBlock, s, a, bias, (const int64_t *)rng_state, y, N, K, grid, rpb,
               kp, groups, relu, thresh, scale);
  else
    GGL_LAUNCH((bias_act_fwd_kernel<1>), grid, kBlock, s, a, bias, (const int64_t *)rng_state, y, N, K, grid, rpb,
               kp, groups, relu, thresh, scale);
  GGL_LAUNCH_CHECK();
  if (thresh) return rng_advance(rng_state, stream);
  return GGL_OK;
}

static inline size_t bwd_partial_bytes(int64_t N, int64_t K) {
  int kp, groups;
  int64_t blocks, rpb;
  geometry(N, K, false, &kp, &groups, &blocks, &rpb);
  size_t a = (size_t)blocks * (size_t)groups;
  geometry(N, K, K % 4 == 0, &kp, &groups, &blocks, &rpb);
  const size_t b = (size_t)blocks * (size_t)groups;
  return ((a > b ? a : b) * (size_t)(K > 0 ? K : 1) * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" size_t ggl_bias_act_bwd_workspace_bytes(int64_t N, int64_t K) {
  const size_t part = bwd_partial_bytes(N, K);
  return part + ggl_colsum_workspace_bytes((int64_t)(part / sizeof(float) / (size_t)(K > 0 ? K : 1)), K);
}

extern "C" int ggl_bias_act_bwd(const float *g, const float *y, int64_t N, int64_t K, int relu,
                                float p_drop, const int64_t *rng_used, float *ga, float *gbias,
                                void *workspace, size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(N >= 0 && K >= 0 && p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "bad arguments");
  if (K == 0) return GGL_OK;
  GGL_REQUIRE((g && ga) || N == 0, GGL_EINVAL, "NULL pointer");
  // rng_used = the {seed, offset} the forward read (a copy taken before it advanced): the dropout mask is
  // redrawn exactly.  Without it the mask is rebuilt from y (y > 0 with ReLU; y != 0 without — exact except
  // for kept activations that are exactly 0).
  // With ReLU the mask read off y is already exact (y > 0 <=> kept and positive; a kept activation that is
  // exactly 0 has zero ReLU gradient anyway), so the redraw — 10 Philox rounds per 4 elements, measured
  // 1.10 -> 1.87 ms on [2.45 M, 256] — is only paid where it is needed: dropout without ReLU.
  const bool redraw = p_drop > 0.0f && rng_used != nullptr && !relu;
  const int masked = relu ? 1 : (redraw ? 3 : (p_drop > 0.0f ? 2 : 0));
  const uint32_t thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  GGL_REQUIRE(!masked || masked == 3 || y || N == 0, GGL_EINVAL, "y is needed to rebuild the ReLU/dropout mask");
  GGL_REQUIRE(!gbias || (workspace && workspace_bytes >= ggl_bias_act_bwd_workspace_bytes(N, K)),
              GGL_EWORKSPACE, "bias_act_bwd workspace too small");
  const bool vec4 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(ga) & 15u) == 0) &&
                    (!masked || masked == 3 || (reinterpret_cast<uintptr_t>(y) & 15u) == 0);
  int kp, groups;
  int64_t blocks, rpb;
  geometry(N, K, vec4, &kp, &groups, &blocks, &rpb);
  const float scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  hipStream_t s = as_stream(stream);
  float *partial = gbias ? static_cast<float *>(workspace) : nullptr;
#define GGL_BAB(V, M)                                                                                    \
  GGL_LAUNCH((bias_act_bwd_kernel<V, M>), blocks, kBlock, s, g, y, ga, N, K, blocks, rpb, kp, groups, scale, \
             redraw ? rng_used : nullptr, thresh, partial)
  if (vec4) {
    if (masked == 0) GGL_BAB(4, 0); else if (masked == 1) GGL_BAB(4, 1); else if (masked == 2) GGL_BAB(4, 2); else GGL_BAB(4, 3);
  } else {
    if (masked == 0) GGL_BAB(1, 0); else if (masked == 1) GGL_BAB(1, 1); else if (masked == 2) GGL_BAB(1, 2); else GGL_BAB(1, 3);
  }
#undef GGL_BAB
  GGL_LAUNCH_CHECK();
  if (gbias) {  // second stage: column sums of the [P, K] partial matrix
    const size_t part = bwd_partial_bytes(N, K);
    return ggl_colsum_f32(partial, blocks * groups, K, gbias, static_cast<char *>(workspace) + part,
                          workspace_bytes - part, stream);
  }
  return GGL_OK;
}
