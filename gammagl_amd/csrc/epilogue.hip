// gammagl_amd/csrc/epilogue.hip — the elementwise step right after every aggregate:
//   GCNConv: out += bias (gcn_conv.py:105-106); GCNModel: relu -> dropout (models/gcn.py:55-59);
//   SAGEConv: + bias, activation (sage_conv.py:102-106).
// torch runs this as add, clamp, dropout (+ a bool mask) forward and masked_scale, threshold, sum(0)
// backward: six passes over [N,K] per hidden layer (measured 4.0 ms of a 110 ms products-sized step,
// profiles/r1_bench_products_summary.txt).  Here it is one pass each way:
//   forward : y = keep(i) * relu(a + bias) / (1 - p)        keep(i) from Philox4x32-10(seed, offset; i)
//   backward: ga = (y > 0) ? g / (1 - p) : 0  (y == 0 exactly where ReLU or dropout killed the value,
//             so no mask tensor is stored), and the bias gradient = column sums of ga accumulated in
//             the same pass (two deterministic stages, as in ggl_colsum_f32).
// The RNG state (seed, offset) lives in device memory and is advanced by a one-thread kernel after every
// forward, so a captured hipGraph draws a fresh mask on each replay.  No LDS, no atomics.
#include "common.hpp"

namespace ggl {

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

// one thread per group of 4 consecutive elements (flat index 4*v .. 4*v+3)
__global__ __launch_bounds__(kBlock) void bias_act_fwd_kernel(const float *__restrict__ a,
                                                              const float *__restrict__ bias,
                                                              const int64_t *__restrict__ rng,
                                                              float *__restrict__ y, int64_t total,
                                                              int64_t K, int relu, uint32_t drop_thresh,
                                                              float scale) {
  const int64_t nvec = (total + 3) >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t seed = drop_thresh ? (uint64_t)rng[0] : 0, offset = drop_thresh ? (uint64_t)rng[1] : 0;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    uint32_t r[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (drop_thresh) {
      const U4 u = philox4x32_10((uint64_t)v, offset, seed);
      r[0] = u.x; r[1] = u.y; r[2] = u.z; r[3] = u.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = 4 * v + j;
      if (i >= total) break;
      float t = a[i];
      if (bias) t = __fadd_rn(t, bias[i % K]);
      if (relu) t = (t < 0.0f) ? 0.0f : t;  // NaN stays NaN, as torch.relu / clamp_min(0)
      if (drop_thresh) t = (r[j] >= drop_thresh) ? __fmul_rn(t, scale) : 0.0f;  // keep with prob 1 - p
      y[i] = t;
    }
  }
}

__global__ void rng_advance_kernel(int64_t *rng, int64_t inc) {
  if (blockIdx.x == 0 && threadIdx.x == 0) rng[1] += inc;
}

// backward + bias-gradient stage 1: same geometry as colsum_stage1_kernel (backward.hip)
__global__ __launch_bounds__(kBlock) void bias_act_bwd_kernel(const float *__restrict__ g,
                                                              const float *__restrict__ y,
                                                              float *__restrict__ ga, int64_t N,
                                                              int64_t K, int64_t rows_per_block, int kp,
                                                              int groups, int masked, float scale,
                                                              float *__restrict__ partial) {
  const int j = threadIdx.x / kp;
  const int k0 = threadIdx.x - j * kp;
  if (j >= groups) return;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  for (int64_t k = k0; k < K; k += kp) {
    float acc = 0.0f;
    for (int64_t r = r0 + j; r < r1; r += groups) {
      const int64_t i = r * K + k;
      float v = g[i];
      if (masked) v = (y[i] > 0.0f) ? __fmul_rn(v, scale) : 0.0f;
      ga[i] = v;
      acc = __fadd_rn(acc, v);
    }
    if (partial) partial[((int64_t)blockIdx.x * groups + j) * K + k] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void bias_colsum_stage2_kernel(const float *__restrict__ partial,
                                                                    int64_t P, int64_t K,
                                                                    float *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float a = 0.f;
  for (int64_t p = 0; p < P; ++p) a = __fadd_rn(a, partial[p * K + k]);
  out[k] = a;
}

static inline void geometry(int64_t N, int64_t K, int *kp, int *groups, int64_t *blocks, int64_t *rpb) {
  *kp = (int)(K < kBlock ? (K > 0 ? K : 1) : kBlock);
  *groups = kBlock / *kp;
  int64_t b = ceil_div(N > 0 ? N : 1, (int64_t)*groups * 16);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  *blocks = b;
  *rpb = ceil_div(N > 0 ? N : 1, b);
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
  const int64_t nvec = (total + 3) / 4;
  int64_t grid = ceil_div(nvec, kBlock);
  if (grid > 8192) grid = 8192;
  GGL_LAUNCH((bias_act_fwd_kernel), grid, kBlock, s, a, bias, (const int64_t *)rng_state, y, total, K,
             relu, thresh, scale);
  GGL_LAUNCH_CHECK();
  if (thresh) {
    GGL_LAUNCH((rng_advance_kernel), 1, 64, s, rng_state, (int64_t)1);
    GGL_LAUNCH_CHECK();
  }
  return GGL_OK;
}

extern "C" size_t ggl_bias_act_bwd_workspace_bytes(int64_t N, int64_t K) {
  int kp, groups;
  int64_t blocks, rpb;
  geometry(N, K, &kp, &groups, &blocks, &rpb);
  return (size_t)blocks * (size_t)groups * (size_t)(K > 0 ? K : 1) * sizeof(float);
}

extern "C" int ggl_bias_act_bwd(const float *g, const float *y, int64_t N, int64_t K, int relu,
                                float p_drop, float *ga, float *gbias, void *workspace,
                                size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(N >= 0 && K >= 0 && p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "bad arguments");
  if (K == 0) return GGL_OK;
  GGL_REQUIRE((g && ga) || N == 0, GGL_EINVAL, "NULL pointer");
  const int masked = (relu || p_drop > 0.0f) ? 1 : 0;
  GGL_REQUIRE(!masked || y || N == 0, GGL_EINVAL, "y is needed to rebuild the ReLU/dropout mask");
  GGL_REQUIRE(!gbias || (workspace && workspace_bytes >= ggl_bias_act_bwd_workspace_bytes(N, K)),
              GGL_EWORKSPACE, "bias_act_bwd workspace too small");
  int kp, groups;
  int64_t blocks, rpb;
  geometry(N, K, &kp, &groups, &blocks, &rpb);
  const float scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  hipStream_t s = as_stream(stream);
  float *partial = gbias ? static_cast<float *>(workspace) : nullptr;
  GGL_LAUNCH((bias_act_bwd_kernel), blocks, kBlock, s, g, y, ga, N, K, rpb, kp, groups, masked, scale,
             partial);
  GGL_LAUNCH_CHECK();
  if (gbias) {
    GGL_LAUNCH((bias_colsum_stage2_kernel), ceil_div(K, kBlock), kBlock, s, (const float *)partial,
               blocks * groups, K, gbias);
    GGL_LAUNCH_CHECK();
  }
  return GGL_OK;
}
