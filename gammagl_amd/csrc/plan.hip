// gammagl_amd/csrc/plan.hip — COO ids -> destination-sorted plan (perm, rowptr, long-row lists), plus
// the library's small housekeeping entry points (errors, options, device info).
//
// This is the step the reference does implicitly on every call by scattering with atomics
// (cuda/segment_sum_cuda.cu:19-31) — and, on the CUDA path, with two device->host syncs per call
// (segment_sum_cuda.cu:51-53).  Here it happens once per edge list: a stable LSD radix sort of
// (id, e) pairs (rocPRIM), a binary search per segment for rowptr, and a two-level ordered
// compaction of the rows longer than `chunk`.  The hot kernels (reduce.hip) never sort, never
// sync and never use atomics.
#include "common.hpp"

#include <cstdarg>

#ifndef GGL_EMULATE
#include <rocprim/rocprim.hpp>
#else
#include <algorithm>
#include <numeric>
#include <vector>
#endif

namespace ggl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int64_t env_i64(const char *name, int64_t dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoll(v) : dflt;
}

int64_t max_grid_x() { return options().max_grid_x; }

Options &options() {
  static Options o = [] {
    Options t;
    t.unroll = env_i64("GGL_UNROLL", t.unroll);
    t.unroll_narrow = env_i64("GGL_UNROLL_NARROW", t.unroll_narrow);
    t.unroll_narrow_max = env_i64("GGL_UNROLL_NARROW_MAX", t.unroll_narrow_max);
    t.xcd_swizzle = env_i64("GGL_XCD_SWIZZLE", t.xcd_swizzle);
    t.force_generic = env_i64("GGL_FORCE_GENERIC", t.force_generic);
    t.ragged4 = env_i64("GGL_RAGGED4", t.ragged4);
    t.col_block = env_i64("GGL_COL_BLOCK", t.col_block);
    t.col_block_min_edges = env_i64("GGL_COL_BLOCK_MIN_EDGES", t.col_block_min_edges);
    t.col_block_min_degree = env_i64("GGL_COL_BLOCK_MIN_DEGREE", t.col_block_min_degree);
    t.row_order = env_i64("GGL_ROW_ORDER", t.row_order);
    t.max_grid_x = env_i64("GGL_MAX_GRID_X", t.max_grid_x);
    t.exact_long_rows = env_i64("GGL_EXACT_LONG_ROWS", t.exact_long_rows);
    t.exact_side_stream = env_i64("GGL_EXACT_SIDE_STREAM", t.exact_side_stream);
    t.exact_long_max = env_i64("GGL_EXACT_LONG_MAX", t.exact_long_max);
    t.hub_one_launch = env_i64("GGL_HUB_ONE_LAUNCH", t.hub_one_launch);
    t.gat_sh_waves = env_i64("GGL_GAT_SH_WAVES", t.gat_sh_waves);
    t.gat_sh_zlds = env_i64("GGL_GAT_SH_ZLDS", t.gat_sh_zlds);
    t.gat_sh_pk = env_i64("GGL_GAT_SH_PK", t.gat_sh_pk);
    t.gat_sh_pipe = env_i64("GGL_GAT_SH_PIPE", t.gat_sh_pipe);
    t.gat_sh_prefetch = env_i64("GGL_GAT_SH_PREFETCH", t.gat_sh_prefetch);
    t.gat_sh_glds = env_i64("GGL_GAT_SH_GLDS", t.gat_sh_glds);
    t.hub_pipe = env_i64("GGL_HUB_PIPE", t.hub_pipe);
    t.hub_priority = env_i64("GGL_HUB_PRIORITY", t.hub_priority);
    t.hop_fused_scans = env_i64("GGL_HOP_FUSED_SCANS", t.hop_fused_scans);
    t.hop_small_scans = env_i64("GGL_HOP_SMALL_SCANS", t.hop_small_scans);
    t.maxbwd_arg32 = env_i64("GGL_MAXBWD_ARG32", t.maxbwd_arg32);
    t.maxbwd_mask = env_i64("GGL_MAXBWD_MASK", t.maxbwd_mask);
    t.maxbwd_mask_kmax = env_i64("GGL_MAXBWD_MASK_KMAX", t.maxbwd_mask_kmax);
    t.maxbwd_mask_scatter = env_i64("GGL_MAXBWD_MASK_SCATTER", t.maxbwd_mask_scatter);
    t.maxbwd_mask_wlane = env_i64("GGL_MAXBWD_MASK_WLANE", t.maxbwd_mask_wlane);
    t.maxbwd_mask_cols = env_i64("GGL_MAXBWD_MASK_COLS", t.maxbwd_mask_cols);
    return t;
  }();
  return o;
}

// flags[0] = some id out of [0, N); flags[1] = ids not non-decreasing
__global__ __launch_bounds__(kBlock) void check_ids_kernel(const int64_t *ids, int64_t E, int64_t N,
                                                           int32_t *flags) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < E; i += stride) {
    const int64_t v = ids[i];
    if (v < 0 || v >= N) flags[0] = 1;  // benign race: every writer stores the same value
    if (i > 0 && ids[i - 1] > v) flags[1] = 1;
  }
}

// the two kernels above in one pass (small plans): flags as check_ids_kernel, keys = ids, vals = 0..E-1
__global__ __launch_bounds__(kBlock) void check_keys_kernel(const int64_t *ids, int64_t E, int64_t N, int32_t *flags,
                                                            uint32_t *keys, int32_t *vals) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < E; i += stride) {
    const int64_t v = ids[i];
    if (v < 0 || v >= N) flags[0] = 1;
    if (i > 0 && ids[i - 1] > v) flags[1] = 1;
    keys[i] = (uint32_t)v;
    vals[i] = (int32_t)i;
  }
}

// out[0] = max_s (rowptr[s + 1] - rowptr[s])   (out zeroed by the caller; E < 2^31 so a row length fits an int)
__global__ __launch_bounds__(kBlock) void max_len_kernel(const int64_t *rowptr, int64_t N, int32_t *out) {
  const int64_t stride = grid_threads();
  int32_t m = 0;
  for (int64_t i = thread_id(); i < N; i += stride) {
    const int32_t len = (int32_t)(rowptr[i + 1] - rowptr[i]);
    m = len > m ? len : m;
  }
  if (m > 0) atomicMax(out, m);   // order-independent
}

__global__ __launch_bounds__(kBlock) void keys_iota_kernel(const int64_t *ids, int64_t E,
                                                           uint32_t *keys, int32_t *vals) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < E; i += stride) {
    if (keys) keys[i] = (uint32_t)ids[i];
    vals[i] = (int32_t)i;
  }
}

// rowptr[s] = first position p with key[p] >= s   (keys sorted ascending), s = 0..N
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void rowptr_kernel(const KeyT *keys, int64_t E, int64_t N,
                                                        int64_t *rowptr) {
  const int64_t stride = grid_threads();
  for (int64_t s = thread_id(); s <= N; s += stride) {
    int64_t lo = 0, hi = E;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < s) lo = mid + 1; else hi = mid;
    }
    rowptr[s] = lo;
  }
}

// ---- two-level ordered scan over rows (no atomics, no LDS) --------------------------------------
constexpr int64_t kSpan = 2048;  // rows per scanning thread
constexpr int64_t kSmallPlan = (int64_t)1 << 22;  // plans up to this many elements are built behind one host read

// per span: number of long rows, number of chunks, longest row
__global__ __launch_bounds__(kBlock) void span_count_kernel(const int64_t *rowptr, int64_t N,
                                                            int64_t chunk, int64_t nspans,
                                                            int64_t *span_long, int64_t *span_chunks,
                                                            int64_t *span_maxlen) {
  const int64_t t = thread_id();
  if (t >= nspans) return;
  const int64_t r0 = t * kSpan, r1 = (r0 + kSpan < N) ? r0 + kSpan : N;
  int64_t nl = 0, nc = 0, mx = 0;
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t len = rowptr[r + 1] - rowptr[r];
    if (len > mx) mx = len;
    if (len > chunk) {
      ++nl;
      nc += (len + chunk - 1) / chunk;
    }
  }
  span_long[t] = nl;
  span_chunks[t] = nc;
  span_maxlen[t] = mx;
}

// one thread: exclusive scan of the span counts in place; totals[0..2] = n_long, n_chunks, max_len
__global__ void span_scan_kernel(int64_t nspans, int64_t *span_long, int64_t *span_chunks,
                                 const int64_t *span_maxlen, int64_t *totals) {
  if (block_id() != 0 || threadIdx.x != 0) return;
  int64_t al = 0, ac = 0, mx = 0;
  for (int64_t t = 0; t < nspans; ++t) {
    const int64_t l = span_long[t], c = span_chunks[t];
    span_long[t] = al;
    span_chunks[t] = ac;
    al += l;
    ac += c;
    if (span_maxlen[t] > mx) mx = span_maxlen[t];
  }
  totals[0] = al;
  totals[1] = ac;
  totals[2] = mx;
}

__global__ __launch_bounds__(kBlock) void span_fill_kernel(const int64_t *rowptr, int64_t N,
                                                           int64_t chunk, int64_t nspans,
                                                           const int64_t *span_long,
                                                           const int64_t *span_chunks,
                                                           int32_t *long_rows, int64_t *chunk_ptr,
                                                           int64_t n_long, int64_t n_chunks) {
  const int64_t t = thread_id();
  if (t == 0) chunk_ptr[n_long] = n_chunks;
  if (t >= nspans) return;
  const int64_t r0 = t * kSpan, r1 = (r0 + kSpan < N) ? r0 + kSpan : N;
  int64_t slot = span_long[t], cacc = span_chunks[t];
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t len = rowptr[r + 1] - rowptr[r];
    if (len > chunk) {
      long_rows[slot] = (int32_t)r;
      chunk_ptr[slot] = cacc;
      ++slot;
      cacc += (len + chunk - 1) / chunk;
    }
  }
}

__global__ __launch_bounds__(kBlock) void gather_i64_i32_kernel(const int64_t *src,
                                                                const int32_t *perm, int64_t E,
                                                                int32_t *out) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < E; i += stride)
    out[i] = (int32_t)src[perm ? (int64_t)perm[i] : i];
}

__global__ __launch_bounds__(kBlock) void gather_rows_f32_kernel(const float *src,
                                                                 const int32_t *perm, int64_t total,
                                                                 int64_t H, float *out) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < total; i += stride) {
    const int64_t p = i / H, h = i - p * H;
    out[i] = src[(perm ? (int64_t)perm[p] : p) * H + h];
  }
}

// out[i, 0:K] = src[idx[i], 0:K] for row-major matrices with row strides (elements) src_ld / out_ld: the send
// buffer of one feature-column block of the halo exchange, gathered straight out of the activation matrix
// (no strided view + index_select + contiguous copy).  VEC = 4: one 16-byte access per lane, K / 4 lanes per
// row, rows coalesced.
template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_ex_kernel(const float *__restrict__ src, int64_t src_ld,
                                                                const int64_t *__restrict__ idx, int64_t n,
                                                                int64_t K, float *__restrict__ out,
                                                                int64_t out_ld) {
  const int64_t kv = K / VEC;
  const int64_t total = n * kv;
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < total; i += stride) {
    const int64_t r = i / kv, c = (i - r * kv) * VEC;
    const float *s = src + idx[r] * src_ld + c;
    float *o = out + r * out_ld + c;
    if (VEC == 4) {
      *reinterpret_cast<float4 *>(o) = *reinterpret_cast<const float4 *>(s);
    } else {
      o[0] = s[0];
    }
  }
}

static inline int64_t grid_for(int64_t n) {
  int64_t g = ceil_div(n, kBlock);
  const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks: grid-stride the rest (guide G11)
  if (g > cap) g = cap;
  return g < 1 ? 1 : g;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int key_bits(int64_t N) {
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < N) ++b;
  return b;
}

#ifndef GGL_EMULATE
static size_t sort_temp_bytes(int64_t E, int bits) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)E, 0u,
                                  (unsigned)bits, (hipStream_t)0);
  return tmp;
}
#endif

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_abi_version(void) { return GGL_ABI_VERSION; }
extern "C" const char *ggl_last_error(void) { return g_err; }

extern "C" int ggl_device_info(int *cus_host, int *wave_host, char *arch_host, int arch_len) {
#ifdef GGL_EMULATE
  if (cus_host) *cus_host = 1;
  if (wave_host) *wave_host = 64;
  if (arch_host && arch_len > 0) snprintf(arch_host, (size_t)arch_len, "emulated");
  return GGL_OK;
#else
  int dev = 0;
  GGL_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  GGL_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  if (cus_host) *cus_host = prop.multiProcessorCount;
  if (wave_host) *wave_host = prop.warpSize;
  if (arch_host && arch_len > 0) snprintf(arch_host, (size_t)arch_len, "%s", prop.gcnArchName);
  return GGL_OK;
#endif
}

extern "C" int ggl_set_option(const char *name, int64_t value) {
  Options &o = options();
  if (!strcmp(name, "unroll")) o.unroll = value;
  else if (!strcmp(name, "unroll_narrow")) o.unroll_narrow = value;
  else if (!strcmp(name, "unroll_narrow_max")) o.unroll_narrow_max = value;
  else if (!strcmp(name, "xcd_swizzle")) o.xcd_swizzle = value;
  else if (!strcmp(name, "force_generic")) o.force_generic = value;
  else if (!strcmp(name, "ragged4")) o.ragged4 = value;
  else if (!strcmp(name, "ragged_max")) o.ragged_max = value;
  else if (!strcmp(name, "col_block")) o.col_block = value;
  else if (!strcmp(name, "col_block_min_edges")) o.col_block_min_edges = value;
  else if (!strcmp(name, "col_block_min_degree")) o.col_block_min_degree = value;
  else if (!strcmp(name, "row_order")) o.row_order = value;
  else if (!strcmp(name, "max_grid_x")) o.max_grid_x = value > 0 ? value : 1;
  else if (!strcmp(name, "exact_long_rows")) o.exact_long_rows = value;
  else if (!strcmp(name, "exact_side_stream")) o.exact_side_stream = value;
  else if (!strcmp(name, "exact_long_max")) o.exact_long_max = value;
  else if (!strcmp(name, "hub_one_launch")) o.hub_one_launch = value;
  else if (!strcmp(name, "gat_sh_waves")) o.gat_sh_waves = value;
  else if (!strcmp(name, "gat_sh_zlds")) o.gat_sh_zlds = value;
  else if (!strcmp(name, "gat_sh_pk")) o.gat_sh_pk = value;
  else if (!strcmp(name, "gat_sh_pipe")) o.gat_sh_pipe = value;
  else if (!strcmp(name, "gat_sh_prefetch")) o.gat_sh_prefetch = value;
  else if (!strcmp(name, "gat_sh_glds")) o.gat_sh_glds = value;
  else if (!strcmp(name, "hub_pipe")) o.hub_pipe = value;
  else if (!strcmp(name, "hub_priority")) o.hub_priority = value;
  else if (!strcmp(name, "hop_fused_scans")) o.hop_fused_scans = value;
  else if (!strcmp(name, "hop_small_scans")) o.hop_small_scans = value;
  else if (!strcmp(name, "maxbwd_arg32")) o.maxbwd_arg32 = value;
  else if (!strcmp(name, "maxbwd_mask")) o.maxbwd_mask = value;
  else if (!strcmp(name, "maxbwd_mask_kmax")) o.maxbwd_mask_kmax = value;
  else if (!strcmp(name, "maxbwd_mask_scatter")) o.maxbwd_mask_scatter = value;
  else if (!strcmp(name, "maxbwd_mask_wlane")) o.maxbwd_mask_wlane = value;
  else if (!strcmp(name, "maxbwd_mask_cols")) o.maxbwd_mask_cols = value;
  else { set_error("unknown option %s", name); return GGL_EINVAL; }
  return GGL_OK;
}

extern "C" int64_t ggl_get_option(const char *name) {
  Options &o = options();
  if (!strcmp(name, "unroll")) return o.unroll;
  if (!strcmp(name, "unroll_narrow")) return o.unroll_narrow;
  if (!strcmp(name, "unroll_narrow_max")) return o.unroll_narrow_max;
  if (!strcmp(name, "xcd_swizzle")) return o.xcd_swizzle;
  if (!strcmp(name, "force_generic")) return o.force_generic;
  if (!strcmp(name, "ragged4")) return o.ragged4;
  if (!strcmp(name, "ragged_max")) return o.ragged_max;
  if (!strcmp(name, "col_block")) return o.col_block;
  if (!strcmp(name, "col_block_min_edges")) return o.col_block_min_edges;
  if (!strcmp(name, "col_block_min_degree")) return o.col_block_min_degree;
  if (!strcmp(name, "row_order")) return o.row_order;
  if (!strcmp(name, "max_grid_x")) return o.max_grid_x;
  if (!strcmp(name, "exact_long_rows")) return o.exact_long_rows;
  if (!strcmp(name, "exact_side_stream")) return o.exact_side_stream;
  if (!strcmp(name, "exact_long_max")) return o.exact_long_max;
  if (!strcmp(name, "hub_one_launch")) return o.hub_one_launch;
  if (!strcmp(name, "gat_sh_waves")) return o.gat_sh_waves;
  if (!strcmp(name, "gat_sh_zlds")) return o.gat_sh_zlds;
  if (!strcmp(name, "gat_sh_pk")) return o.gat_sh_pk;
  if (!strcmp(name, "gat_sh_pipe")) return o.gat_sh_pipe;
  if (!strcmp(name, "gat_sh_prefetch")) return o.gat_sh_prefetch;
  if (!strcmp(name, "gat_sh_glds")) return o.gat_sh_glds;
  if (!strcmp(name, "hub_pipe")) return o.hub_pipe;
  if (!strcmp(name, "hub_priority")) return o.hub_priority;
  if (!strcmp(name, "hop_fused_scans")) return o.hop_fused_scans;
  if (!strcmp(name, "hop_small_scans")) return o.hop_small_scans;
  if (!strcmp(name, "maxbwd_arg32")) return o.maxbwd_arg32;
  if (!strcmp(name, "maxbwd_mask")) return o.maxbwd_mask;
  if (!strcmp(name, "maxbwd_mask_kmax")) return o.maxbwd_mask_kmax;
  if (!strcmp(name, "maxbwd_mask_scatter")) return o.maxbwd_mask_scatter;
  if (!strcmp(name, "maxbwd_mask_wlane")) return o.maxbwd_mask_wlane;
  if (!strcmp(name, "maxbwd_mask_cols")) return o.maxbwd_mask_cols;
  return -1;
}

// ---- host policy, one copy (see the header) ----------------------------------------------------------------------------
extern "C" int64_t ggl_policy_chunk(int64_t E) {
  static const int64_t forced = env_i64("GGL_LONG_ROW", 0);
  if (forced > 0) return forced;
  int64_t c = 4096;
  while (c > 256 && c * (256 * 32) > E) c >>= 1;
  return c;
}
extern "C" int64_t ggl_policy_spmm_width(int reduce, int64_t K, int64_t E, int64_t N_in) {
  if (K <= 0 || E < 8 * N_in) return K;
  if (reduce == 0) {
    if (K > 256 && K % 64 != 0) return K + (64 - K % 64);
    if (K % 4 != 0 && K >= 8) return K + (4 - K % 4);
    return K;
  }
  if (K > 128 && K % 4 != 0) return K + (4 - K % 4);
  return K;
}
extern "C" int64_t ggl_policy_head_channels(int64_t C, int64_t E, int64_t N_in) {
  return (C % 4 != 0 && C >= 8 && E >= 8 * N_in) ? C + (4 - C % 4) : C;
}
// (... and only on edge lists whose walk dominates its launches: on a sampled block the four elementwise kernels of the
//  prescale — counts, clamp, cast, divide — cost more than the per-edge degree lookup they save; round 5)
extern "C" int ggl_policy_mean_bwd_prescale(int64_t E, int64_t N_in) { return (E >= 4 * N_in && E >= ((int64_t)1 << 22)) ? 1 : 0; }
extern "C" int ggl_policy_gradw_sorted(int64_t H, int64_t C) {
  return (C % 4 == 0 && (C > 16 || (C > 8 && H * C >= 256))) ? 1 : 0;
}
extern "C" int64_t ggl_policy_xcd_run_rows(int64_t E, double locality) {
  static const int64_t forced = env_i64("GGL_XCD_RUN_ROWS", -1);
  if (forced >= 0) return forced;
  return (E >= ((int64_t)1 << 22) && locality > 0.5) ? 2048 : 0;
}
extern "C" int ggl_policy_row_order(int64_t *window_host, int64_t *heavy_host) {
  static const int64_t window = env_i64("GGL_ROW_ORDER_WINDOW", 2048);
  if (window_host) *window_host = window;
  if (heavy_host) *heavy_host = 1024;
  return GGL_OK;
}

extern "C" size_t ggl_plan_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return 0;
  size_t b = 256;                                        // flags
  b += 2 * align_up((size_t)E * 4, 256);                 // keys in / out
  b += align_up((size_t)E * 4, 256);                     // vals in
#ifndef GGL_EMULATE
  b += align_up(sort_temp_bytes(E > 0 ? E : 1, key_bits(N)), 256);
#endif
  b += 3 * align_up((size_t)(ceil_div(N > 0 ? N : 1, kSpan)) * 8, 256) + 256;  // max_len scan
  return b;
}

extern "C" size_t ggl_plan_long_workspace_bytes(int64_t N) {
  return 3 * align_up((size_t)(ceil_div(N > 0 ? N : 1, kSpan)) * 8, 256) + 256;
}

static int run_span_scan(const int64_t *rowptr, int64_t N, int64_t chunk, char *ws, hipStream_t s,
                         int64_t **span_long, int64_t **span_chunks, int64_t totals_host[3]) {
  const int64_t nspans = ceil_div(N > 0 ? N : 1, kSpan);
  const size_t seg = align_up((size_t)nspans * 8, 256);
  int64_t *sl = reinterpret_cast<int64_t *>(ws);
  int64_t *sc = reinterpret_cast<int64_t *>(ws + seg);
  int64_t *sm = reinterpret_cast<int64_t *>(ws + 2 * seg);
  int64_t *tot = reinterpret_cast<int64_t *>(ws + 3 * seg);
  GGL_LAUNCH((span_count_kernel), ceil_div(nspans, kBlock), kBlock, s, rowptr, N, chunk, nspans, sl,
             sc, sm);
  GGL_LAUNCH_CHECK();
  GGL_LAUNCH((span_scan_kernel), 1, 64, s, nspans, sl, sc, sm, tot);
  GGL_LAUNCH_CHECK();
  GGL_HIP_CHECK(hipMemcpyAsync(totals_host, tot, 24, hipMemcpyDeviceToHost, s));
  GGL_HIP_CHECK(hipStreamSynchronize(s));
  *span_long = sl;
  *span_chunks = sc;
  return GGL_OK;
}

extern "C" int ggl_plan_build(const int64_t *ids, int64_t E, int64_t N, int32_t *perm,
                              int64_t *rowptr, void *workspace, size_t workspace_bytes,
                              void *stream, int32_t *is_sorted_host, int64_t *max_len_host) {
  GGL_REQUIRE(E >= 0 && N >= 0, GGL_EINVAL, "negative size");
  GGL_REQUIRE(E < ((int64_t)1 << 31), GGL_EINVAL, "E >= 2^31 elements per plan is not supported");
  GGL_REQUIRE(N < ((int64_t)1 << 31), GGL_EINVAL, "N >= 2^31 segments is not supported");
  GGL_REQUIRE(rowptr != nullptr, GGL_EINVAL, "rowptr is NULL");
  GGL_REQUIRE((ids && perm) || E == 0, GGL_EINVAL, "ids/perm is NULL");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_plan_workspace_bytes(E, N), GGL_EWORKSPACE,
              "plan workspace too small: need %zu bytes", ggl_plan_workspace_bytes(E, N));
  hipStream_t s = as_stream(stream);
  char *ws = static_cast<char *>(workspace);
  int32_t *flags = reinterpret_cast<int32_t *>(ws);
  size_t off = 256;
  uint32_t *keys_in = reinterpret_cast<uint32_t *>(ws + off);
  off += align_up((size_t)E * 4, 256);
  uint32_t *keys_out = reinterpret_cast<uint32_t *>(ws + off);
  off += align_up((size_t)E * 4, 256);
  int32_t *vals_in = reinterpret_cast<int32_t *>(ws + off);
  off += align_up((size_t)E * 4, 256);

  int32_t flags_host[4] = {0, 0, 0, 0};
  GGL_HIP_CHECK(hipMemsetAsync(flags, 0, 256, s));
  if (E > 0 && E <= kSmallPlan) {
    // A small plan is usually a FRESH one (a sampled block's edge list, rebuilt every mini-batch by loaders that hand
    // out COO): its cost is host round trips, not device work.  Everything is queued behind ONE read at the end —
    // the ids are sorted whether or not they arrive sorted (a stable sort of sorted keys leaves the identity), range
    // and sortedness flags and the longest row come back together.  (Large plans below keep the early read: skipping
    // the sort of an already sorted 10^8-element list is worth a round trip.)
    GGL_LAUNCH((check_keys_kernel), grid_for(E), kBlock, s, ids, E, N, flags, keys_in, vals_in);
    GGL_LAUNCH_CHECK();
#ifndef GGL_EMULATE
    const int bits = key_bits(N);
    size_t tmp = sort_temp_bytes(E, bits);
    GGL_HIP_CHECK(rocprim::radix_sort_pairs(ws + off, tmp, (const uint32_t *)keys_in, keys_out,
                                            (const int32_t *)vals_in, perm, (size_t)E, 0u, (unsigned)bits, s));
#else
    {
      std::vector<int32_t> order((size_t)E);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return keys_in[a] < keys_in[b]; });
      for (int64_t i = 0; i < E; ++i) {
        perm[i] = order[(size_t)i];
        keys_out[i] = keys_in[order[(size_t)i]];
      }
    }
#endif
    GGL_LAUNCH((rowptr_kernel<uint32_t>), grid_for(N + 1), kBlock, s, (const uint32_t *)keys_out, E, N, rowptr);
    GGL_LAUNCH_CHECK();
    GGL_LAUNCH((max_len_kernel), grid_for(N > 0 ? N : 1), kBlock, s, (const int64_t *)rowptr, N, flags + 2);
    GGL_LAUNCH_CHECK();
    GGL_HIP_CHECK(hipMemcpyAsync(flags_host, flags, 16, hipMemcpyDeviceToHost, s));
    GGL_HIP_CHECK(hipStreamSynchronize(s));
    GGL_REQUIRE(flags_host[0] == 0, GGL_EINDEX, "segment id out of range [0, %lld)", (long long)N);
    if (is_sorted_host) *is_sorted_host = flags_host[1] == 0 ? 1 : 0;
    if (max_len_host) *max_len_host = flags_host[2];
    return GGL_OK;
  }
  if (E > 0) {
    GGL_LAUNCH((check_ids_kernel), grid_for(E), kBlock, s, ids, E, N, flags);
    GGL_LAUNCH_CHECK();
  }
  GGL_HIP_CHECK(hipMemcpyAsync(flags_host, flags, 8, hipMemcpyDeviceToHost, s));
  GGL_HIP_CHECK(hipStreamSynchronize(s));
  GGL_REQUIRE(flags_host[0] == 0, GGL_EINDEX, "segment id out of range [0, %lld)", (long long)N);
  const bool sorted = flags_host[1] == 0;
  if (is_sorted_host) *is_sorted_host = sorted ? 1 : 0;

  if (sorted) {
    if (E > 0) {
      GGL_LAUNCH((keys_iota_kernel), grid_for(E), kBlock, s, ids, E, (uint32_t *)nullptr, perm);
      GGL_LAUNCH_CHECK();
    }
    GGL_LAUNCH((rowptr_kernel<int64_t>), grid_for(N + 1), kBlock, s, ids, E, N, rowptr);
    GGL_LAUNCH_CHECK();
  } else {
    GGL_LAUNCH((keys_iota_kernel), grid_for(E), kBlock, s, ids, E, keys_in, vals_in);
    GGL_LAUNCH_CHECK();
#ifndef GGL_EMULATE
    const int bits = key_bits(N);
    size_t tmp = sort_temp_bytes(E, bits);
    void *tmp_ptr = ws + off;
    off += align_up(tmp, 256);
    GGL_HIP_CHECK(rocprim::radix_sort_pairs(tmp_ptr, tmp, (const uint32_t *)keys_in, keys_out,
                                            (const int32_t *)vals_in, perm, (size_t)E, 0u,
                                            (unsigned)bits, s));
#else
    std::vector<int32_t> order((size_t)E);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int32_t a, int32_t b) { return keys_in[a] < keys_in[b]; });
    for (int64_t i = 0; i < E; ++i) {
      perm[i] = order[(size_t)i];
      keys_out[i] = keys_in[order[(size_t)i]];
    }
#endif
    GGL_LAUNCH((rowptr_kernel<uint32_t>), grid_for(N + 1), kBlock, s, (const uint32_t *)keys_out, E,
               N, rowptr);
    GGL_LAUNCH_CHECK();
  }
  // longest row (host fact used to size the long-row machinery)
  int64_t *sl, *sc, totals[3] = {0, 0, 0};
  char *scan_ws = ws + ggl_plan_workspace_bytes(E, N) - ggl_plan_long_workspace_bytes(N);
  int rc = run_span_scan(rowptr, N, (int64_t)1 << 62, scan_ws, s, &sl, &sc, totals);
  if (rc) return rc;
  if (max_len_host) *max_len_host = totals[2];
  return GGL_OK;
}

extern "C" int ggl_plan_long_count(const int64_t *rowptr, int64_t N, int64_t chunk, void *workspace,
                                   size_t workspace_bytes, void *stream, int64_t *n_long_host,
                                   int64_t *n_chunks_host) {
  GGL_REQUIRE(rowptr && chunk > 0 && N >= 0, GGL_EINVAL, "bad arguments");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_plan_long_workspace_bytes(N), GGL_EWORKSPACE,
              "long-row workspace too small");
  int64_t *sl, *sc, totals[3] = {0, 0, 0};
  int rc = run_span_scan(rowptr, N, chunk, static_cast<char *>(workspace), as_stream(stream), &sl,
                         &sc, totals);
  if (rc) return rc;
  if (n_long_host) *n_long_host = totals[0];
  if (n_chunks_host) *n_chunks_host = totals[1];
  return GGL_OK;
}

extern "C" int ggl_plan_long_fill(const int64_t *rowptr, int64_t N, int64_t chunk, int64_t n_long,
                                  int32_t *long_rows, int64_t *chunk_ptr, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(rowptr && chunk > 0 && N >= 0 && n_long >= 0, GGL_EINVAL, "bad arguments");
  GGL_REQUIRE(chunk_ptr && (long_rows || n_long == 0), GGL_EINVAL, "long_rows/chunk_ptr is NULL");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_plan_long_workspace_bytes(N), GGL_EWORKSPACE,
              "long-row workspace too small");
  hipStream_t s = as_stream(stream);
  int64_t *sl, *sc, totals[3] = {0, 0, 0};
  int rc = run_span_scan(rowptr, N, chunk, static_cast<char *>(workspace), s, &sl, &sc, totals);
  if (rc) return rc;
  GGL_REQUIRE(totals[0] == n_long, GGL_EINVAL, "n_long mismatch: plan has %lld long rows",
              (long long)totals[0]);
  const int64_t nspans = ceil_div(N > 0 ? N : 1, kSpan);
  GGL_LAUNCH((span_fill_kernel), ceil_div(nspans, kBlock), kBlock, s, rowptr, N, chunk, nspans,
             (const int64_t *)sl, (const int64_t *)sc, long_rows, chunk_ptr, n_long, totals[1]);
  GGL_LAUNCH_CHECK();
  GGL_HIP_CHECK(hipStreamSynchronize(s));
  return GGL_OK;
}

// inv[perm[i]] = i: the inverse of a permutation of [0, n) (GraphPlan.tpos = the inverse of posT: forward sorted position
// -> transposed sorted position, which ggl_spmm_max_mask scatters an edge's winner bits to)
__global__ __launch_bounds__(kBlock) void invert_perm_kernel(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ inv) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < n; i += stride) inv[perm[i]] = (int32_t)i;
}

extern "C" int ggl_invert_perm(const int32_t *perm, int64_t n, int32_t *inv, void *stream) {
  GGL_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && ((perm && inv) || n == 0), GGL_EINVAL, "bad arguments");
  if (n == 0) return GGL_OK;
  GGL_LAUNCH((invert_perm_kernel), grid_for(n), kBlock, as_stream(stream), perm, n, inv);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_gather_i64_to_i32(const int64_t *src, const int32_t *perm, int64_t E,
                                     int32_t *out, void *stream) {
  GGL_REQUIRE(E >= 0 && ((src && out) || E == 0), GGL_EINVAL, "bad arguments");
  if (E == 0) return GGL_OK;
  GGL_LAUNCH((gather_i64_i32_kernel), grid_for(E), kBlock, as_stream(stream), src, perm, E, out);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_gather_rows_f32(const float *src, const int32_t *perm, int64_t E, int64_t H,
                                   float *out, void *stream) {
  GGL_REQUIRE(E >= 0 && H > 0 && ((src && out) || E == 0), GGL_EINVAL, "bad arguments");
  if (E == 0) return GGL_OK;
  GGL_LAUNCH((gather_rows_f32_kernel), grid_for(E * H), kBlock, as_stream(stream), src, perm, E * H,
             H, out);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

// ---- ggl_calib_stream: the streaming yardstick of bench.py's roofline leg ------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void calib_stream_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                              int64_t n, float *__restrict__ sink) {
  const int64_t stride = grid_threads();
  float4 acc{0.0f, 0.0f, 0.0f, 0.0f};
  int64_t i = thread_id();
  for (; i + 7 * stride < n; i += 8 * stride) {     // eight independent 16-byte loads in flight per lane
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 1) dst[i + u * stride] = v[u];
      else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  for (; i < n; i += stride) {
    const float4 a = src[i];
    if (MODE == 1) dst[i] = a; else { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  }
  if (MODE == 0) {   // keep the loads alive: one float per wavefront-sized group of threads
    const float v = acc.x + acc.y + acc.z + acc.w;
    if (v == 1234.5678f || (threadIdx.x & 63) == 0) sink[(thread_id() >> 6) & 65535] = v;
  }
}

extern "C" int ggl_calib_stream(const float *src, float *dst, int64_t n_vec4, int mode, void *stream) {
  GGL_REQUIRE(n_vec4 >= 0 && (mode == 0 || mode == 1), GGL_EINVAL, "bad arguments");
  if (n_vec4 == 0) return GGL_OK;
  GGL_REQUIRE(src && dst && (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0,
              GGL_EINVAL, "src / dst must be 16-byte aligned");
  int64_t g = ceil_div(n_vec4, (int64_t)kBlock * 4);
  if (g > 256 * 8) g = 256 * 8;        // 256 CUs x 8 blocks of 4 wavefronts (full occupancy), grid-stride for the rest
  const float4 *s4 = reinterpret_cast<const float4 *>(src);
  if (mode == 1) GGL_LAUNCH((calib_stream_kernel<1>), g, kBlock, as_stream(stream), s4, reinterpret_cast<float4 *>(dst), n_vec4, dst);
  else GGL_LAUNCH((calib_stream_kernel<0>), g, kBlock, as_stream(stream), s4, reinterpret_cast<float4 *>(dst), n_vec4, dst);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_gather_rows_f32_ex(const float *src, int64_t src_ld, const int64_t *idx, int64_t n, int64_t K,
                                      float *out, int64_t out_ld, void *stream) {
  GGL_REQUIRE(n >= 0 && K >= 0 && src_ld >= K && out_ld >= K, GGL_EINVAL, "bad arguments");
  if (n == 0 || K == 0) return GGL_OK;
  GGL_REQUIRE(src && idx && out, GGL_EINVAL, "NULL pointer");
  const bool vec4 = K % 4 == 0 && src_ld % 4 == 0 && out_ld % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  // enough blocks in flight to cover the random-row latency (a pure gather: 32 blocks per CU)
  int64_t g = ceil_div(n * (vec4 ? K / 4 : K), kBlock);
  if (g > 256 * 32) g = 256 * 32;
  if (vec4)
    GGL_LAUNCH((gather_rows_ex_kernel<4>), g, kBlock, as_stream(stream), src, src_ld, idx, n, K, out, out_ld);
  else
    GGL_LAUNCH((gather_rows_ex_kernel<1>), g, kBlock, as_stream(stream), src, src_ld, idx, n, K, out, out_ld);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

// ------------------------------------------------------------------------------------------------
// Format conversion entry points: the device-side counterparts of
// gammagl/ops/sparse ind2ptr / ptr2ind (cpu/convert.cpp:58-128; numpy fallback ops/sparse/__init__.py:
// 23-41; CUDA: a binary search per output, cuda/convert.cu:41-60,85-104) and of
// utils/sort_edge_index.py:30-44 (argsort of row * N + col).
// ------------------------------------------------------------------------------------------------
namespace ggl {

// ind[p] = r  with ptr[r] <= p < ptr[r+1]
__global__ __launch_bounds__(kBlock) void ptr2ind_kernel(const int64_t *__restrict__ ptr, int64_t M,
                                                         int64_t E, int64_t *__restrict__ ind) {
  const int64_t stride = grid_threads();
  for (int64_t p = thread_id(); p < E; p += stride) {
    int64_t lo = 0, hi = M;  // first r with ptr[r+1] > p
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (ptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
    }
    ind[p] = lo;
  }
}

__global__ __launch_bounds__(kBlock) void edge_keys_kernel(const int64_t *__restrict__ major,
                                                           const int64_t *__restrict__ minor,
                                                           int64_t E, int64_t N,
                                                           uint64_t *__restrict__ keys,
                                                           int32_t *__restrict__ vals) {
  const int64_t stride = grid_threads();
  for (int64_t i = thread_id(); i < E; i += stride) {
    keys[i] = (uint64_t)major[i] * (uint64_t)N + (uint64_t)minor[i];
    vals[i] = (int32_t)i;
  }
}

static inline int key_bits64(int64_t N) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) < N) ++b;
  return 2 * b < 64 ? 2 * b : 64;
}

#ifndef GGL_EMULATE
static size_t sort64_temp_bytes(int64_t E, int bits) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)E, 0u,
                                  (unsigned)bits, (hipStream_t)0);
  return tmp;
}
#endif

}  // namespace ggl

// ptr[M+1] = exclusive prefix of the histogram of ind (ind need not be sorted: the numpy fallback's
// bincount + cumsum).  For sorted ind this is the binary-search form of cuda/convert.cu:41-60.
extern "C" size_t ggl_ind2ptr_workspace_bytes(int64_t E, int64_t M) {
  return ggl_plan_workspace_bytes(E, M) + align_up((size_t)(E > 0 ? E : 1) * 4, 256);
}

extern "C" int ggl_ind2ptr(const int64_t *ind, int64_t E, int64_t M, int64_t *ptr, void *workspace,
                           size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_ind2ptr_workspace_bytes(E, M), GGL_EWORKSPACE,
              "ind2ptr workspace too small");
  char *ws = static_cast<char *>(workspace);
  int32_t *perm = reinterpret_cast<int32_t *>(ws);
  const size_t off = align_up((size_t)(E > 0 ? E : 1) * 4, 256);
  int32_t sorted = 0;
  int64_t max_len = 0;
  return ggl_plan_build(ind, E, M, perm, ptr, ws + off, workspace_bytes - off, stream, &sorted,
                        &max_len);
}

extern "C" int ggl_ptr2ind(const int64_t *ptr, int64_t M, int64_t E, int64_t *ind, void *stream) {
  GGL_REQUIRE(M >= 0 && E >= 0 && (ptr || M == 0) && (ind || E == 0), GGL_EINVAL, "bad arguments");
  if (E == 0) return GGL_OK;
  GGL_LAUNCH((ptr2ind_kernel), grid_for(E), kBlock, as_stream(stream), ptr, M, E, ind);
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

// perm[E] = stable argsort of major[i] * N + minor[i]  (sort_edge_index.py:36-39); ties keep the
// original order, which the reference's (unstable) argsort leaves unspecified.
extern "C" size_t ggl_sort_edges_workspace_bytes(int64_t E, int64_t N) {
  size_t b = 2 * align_up((size_t)(E > 0 ? E : 1) * 8, 256) + align_up((size_t)(E > 0 ? E : 1) * 4, 256);
#ifndef GGL_EMULATE
  b += align_up(sort64_temp_bytes(E > 0 ? E : 1, key_bits64(N)), 256);
#endif
  return b;
}

extern "C" int ggl_sort_edges(const int64_t *major, const int64_t *minor, int64_t E, int64_t N,
                              int32_t *perm, void *workspace, size_t workspace_bytes, void *stream) {
  GGL_REQUIRE(E >= 0 && N >= 0 && E < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), GGL_EINVAL,
              "bad sizes");
  if (E == 0) return GGL_OK;
  GGL_REQUIRE(major && minor && perm, GGL_EINVAL, "NULL pointer");
  GGL_REQUIRE(workspace && workspace_bytes >= ggl_sort_edges_workspace_bytes(E, N), GGL_EWORKSPACE,
              "sort_edges workspace too small");
  hipStream_t s = as_stream(stream);
  char *ws = static_cast<char *>(workspace);
  uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws);
  size_t off = align_up((size_t)E * 8, 256);
  uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws + off);
  off += align_up((size_t)E * 8, 256);
  int32_t *vals_in = reinterpret_cast<int32_t *>(ws + off);
  off += align_up((size_t)E * 4, 256);
  GGL_LAUNCH((edge_keys_kernel), grid_for(E), kBlock, s, major, minor, E, N, keys_in, vals_in);
  GGL_LAUNCH_CHECK();
#ifndef GGL_EMULATE
  const int bits = key_bits64(N);
  size_t tmp = sort64_temp_bytes(E, bits);
  GGL_HIP_CHECK(rocprim::radix_sort_pairs(ws + off, tmp, (const uint64_t *)keys_in, keys_out,
                                          (const int32_t *)vals_in, perm, (size_t)E, 0u,
                                          (unsigned)bits, s));
#else
  std::vector<int32_t> order((size_t)E);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int32_t a, int32_t b) { return keys_in[a] < keys_in[b]; });
  for (int64_t i = 0; i < E; ++i) perm[i] = order[(size_t)i];
  (void)keys_out;
#endif
  return GGL_OK;
}
