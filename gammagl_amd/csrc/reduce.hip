// gammagl_amd/csrc/reduce.hip — the row-reduction kernels behind every aggregate on the path:
//   ggl_segment_{sum,mean,max}          (supersede cpu/segment_*_cpu.cpp, cuda/segment_*_cuda.cu)
//   ggl_spmm_{sum,mean,max}[_bwd]       (supersede cpu/spmm_*_cpu.cpp, cuda/spmm_sum_cuda.cu)
//   ggl_bspmm_sum                       (supersedes cpu/bspmm_sum_cpu.cpp)
//
// One design for all of them (MI355X-first, not a port of the reference's one-thread-per-(edge,k)
// atomicAdd scatter):
//   * rows of the destination-sorted plan are the unit of work; a group of L = 2^logL lanes
//     (L = 64 -> one wavefront per row) owns a row and strides the feature axis, VEC contiguous
//     elements per lane (float4 = 16 B/lane -> 1 KiB per wave-load for K = 256);
//   * the lane group walks the row's elements in ascending original order with U loads in
//     flight, accumulating in order in registers: no atomics, no LDS, no cross-lane traffic,
//     one coalesced store per output row.  Summation order == the reference's serial CPU order,
//     so short rows are bit-identical to it and the argmax tie-break ("first edge wins") is exact;
//   * for L = 64 the row id, rowptr, column indices and edge weights are wave-uniform: they are
//     read through the scalar path (s_load) and broadcast for free, the vector memory pipe only
//     carries feature rows;
//   * rows longer than plan->chunk are cut into chunks reduced by independent wavefronts into a
//     partial buffer and combined in chunk order (deterministic; bounds the tail a power-law hub
//     would otherwise put on one wavefront);
//   * block b -> row block remap keeps each XCD's private L2 on a contiguous row range.
// Roofline: HBM.  Algorithmic bytes per edge = 4K (feature row) + 4 (col) + 4 (weight);
// per output row = 4K + 8.
#include "common.hpp"

namespace ggl {

enum Mode {
  MODE_SEG = 0,        // value = x[e,:],                    e = perm ? perm[p] : p
  MODE_SPMM = 1,       // value = w * x[col[p],:]
  MODE_BSPMM = 2,      // value = w[.,h] * x[col[p],h,:],    h = k / C
  MODE_MEANBWD = 3,    // value = g[col[p],:] / count[col[p]] * w       (spmm_mean_cpu.cpp:95-101)
  MODE_MAXBWD = 4,     // value = w * g[col[p],k] if argsrc[col[p],k] == row (spmm_max_cpu.cpp:88-93)
  MODE_SPMM_EPI = 5,   // MODE_SPMM + the layer epilogue applied to the finished row before its only store:
                       // y = dropout(relu(reduced + add[row] + bias))  (gcn_conv.py:105-106, models/gcn.py:55-59;
                       // add = SAGEConv's fc_self(x_dst) term, sage_conv.py:100-108)
  MODE_SEG_EPI = 6,    // MODE_SEG + the same epilogue (the message() + aggregate() route of a sampled block)
  MODE_MAXBWD32 = 7,   // MODE_MAXBWD reading its witnesses from a compact int32 copy (ggl_spmm_max_bwd32): its own
                       // instantiation — a run-time width switch inside the walk was measured and loses (the backend stops
                       // merging a lane's four witness loads: K = 256 forward + backward 68.0 -> 75.9 ms)
  MODE_MAXBWDM = 8     // MODE_MAXBWD reading ONE BIT per (edge, column) — "this edge's source is the row's witness" — from a
                       // mask in transposed position order (ggl_spmm_max_mask builds it in destination order, where the
                       // witness row is wave-uniform): K / 8 bytes per edge beside the 4K-byte gradient row instead of 8K
};
constexpr bool maxbwd_like(int mode) { return mode == MODE_MAXBWD || mode == MODE_MAXBWD32 || mode == MODE_MAXBWDM; }
constexpr bool spmm_like(int mode) { return mode == MODE_SPMM || mode == MODE_SPMM_EPI; }
constexpr bool seg_like(int mode) { return mode == MODE_SEG || mode == MODE_SEG_EPI; }
constexpr bool epi_mode(int mode) { return mode == MODE_SPMM_EPI || mode == MODE_SEG_EPI; }

// How positions map to element / weight indices, fixed at compile time for the hot f32 kernels so
// the inner loop carries no pointer tests:
//   MODE_SEG : IDX_DIRECT = ids arrived sorted (e = p), IDX_PERM = e = perm[p]
//   others   : IDX_NONE = no weights, IDX_DIRECT = w[p], IDX_PERM = w[perm[p]]
//   IDX_RUNTIME = decide from the pointers at run time (rare modes, non-f32 dtypes)
enum IdxMode { IDX_RUNTIME = -1, IDX_NONE = 0, IDX_DIRECT = 1, IDX_PERM = 2 };

// Scalars of one launch.  Pointers travel as separate `const T* __restrict__` kernel parameters:
// that is what lets the backend prove the index/weight loads are not clobbered by the output
// stores and issue them on the scalar path (s_load) when the row is wave-uniform.
struct ReduceDims {
  int64_t N, K, E;
  int64_t arg_fill;
  int64_t chunk;
  int64_t nblocks;
  int64_t H, C;
  int64_t n_long, n_chunks;
  int64_t chunk_blocks;  // leading blocks of the launch that reduce long-row chunks
  int exact_long;        // long rows arrive as ONE partial each, summed in the reference's serial order (hubf32.hip)
  int logL;
  int swizzle;
  int w_by_pos;
  // MODE_SPMM_EPI: same mask as ggl_bias_act_fwd draws for the same (rng state, N, K) — Philox word of
  // vector (row, k / epi_vec), component k % epi_vec
  // row strides (elements) of x and out, and out += instead of out = : lets a caller aggregate a column
  // block of a wider matrix in place and add a second edge set (halo sources) onto an existing result
  int64_t x_ld, out_ld;
  int accumulate;
  int epi_relu;
  int epi_vec;
  uint32_t epi_thresh;
  float epi_scale;
  // a column block [epi_col0, epi_col0 + K) of a row of epi_K columns: the dropout word of element (row, k)
  // is the one the full-width launch draws, so a result assembled block by block carries the same mask
  int64_t epi_K, epi_col0;
  int64_t add_ld;          // row stride of epi_add
  int64_t mask_words;      // MODE_MAXBWDM: 32-bit words per edge record of the winner mask
  int64_t mask_col0;       // ... and the first column of this launch inside the full row (column-block launches)
};

template <typename S> struct RPtrs {
  const S *__restrict__ x;
  const int32_t *__restrict__ perm;
  const int32_t *__restrict__ col;
  const float *__restrict__ w;
  const int64_t *__restrict__ rowptr;
  const int64_t *__restrict__ aux_rowptr;
  const int64_t *__restrict__ aux_arg;
  const float *__restrict__ epi_bias;
  const int64_t *__restrict__ epi_rng;
  const float *__restrict__ epi_add;
};

// ---- VEC-wide loads / stores of storage elements ------------------------------------------------
template <typename S, int VEC> struct VecIO {
  static __device__ __forceinline__ void load(const S *__restrict__ p, S (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = p[i];
  }
  static __device__ __forceinline__ void store(S *__restrict__ p, const S (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = v[i];
  }
};
template <> struct VecIO<float, 4> {
  static __device__ __forceinline__ void load(const float *__restrict__ p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(p);  // global_load_dwordx4
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float *__restrict__ p, const float (&v)[4]) {
    float4 t;
    t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    *reinterpret_cast<float4 *>(p) = t;
  }
};

// 16-byte accesses for the 16-bit float storage types (8 elements) and for f64 (2 elements)
template <> struct VecIO<uint16_t, 8> {
  static __device__ __forceinline__ void load(const uint16_t *__restrict__ p, uint16_t (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4 *>(p);
    v[0] = (uint16_t)t.x; v[1] = (uint16_t)(t.x >> 16); v[2] = (uint16_t)t.y; v[3] = (uint16_t)(t.y >> 16);
    v[4] = (uint16_t)t.z; v[5] = (uint16_t)(t.z >> 16); v[6] = (uint16_t)t.w; v[7] = (uint16_t)(t.w >> 16);
  }
  static __device__ __forceinline__ void store(uint16_t *__restrict__ p, const uint16_t (&v)[8]) {
    uint4 t;
    t.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); t.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
    t.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); t.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
    *reinterpret_cast<uint4 *>(p) = t;
  }
};
template <> struct VecIO<double, 2> {
  static __device__ __forceinline__ void load(const double *__restrict__ p, double (&v)[2]) {
    const double2 t = *reinterpret_cast<const double2 *>(p);
    v[0] = t.x; v[1] = t.y;
  }
  static __device__ __forceinline__ void store(double *__restrict__ p, const double (&v)[2]) {
    double2 t;
    t.x = v[0]; t.y = v[1];
    *reinterpret_cast<double2 *>(p) = t;
  }
};

// 16-byte accesses for the integer storage types too (int32 x 4, int64 x 2): an [E, 64] int32 message tensor used to
// walk with one dword per lane (the VEC = 1 kernels) at 3.7 TB/s where the f32 rows of the same bytes run at 5.7
template <typename S, int VEC> struct alignas(16) Pack16 { S v[VEC]; };
#define GGL_INT_VECIO(S, VEC)                                                                       \
  template <> struct VecIO<S, VEC> {                                                                \
    static __device__ __forceinline__ void load(const S *__restrict__ p, S (&v)[VEC]) {             \
      const Pack16<S, VEC> t = *reinterpret_cast<const Pack16<S, VEC> *>(p);                        \
      _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = t.v[i];                                \
    }                                                                                               \
    static __device__ __forceinline__ void store(S *__restrict__ p, const S (&v)[VEC]) {            \
      Pack16<S, VEC> t;                                                                             \
      _Pragma("unroll") for (int i = 0; i < VEC; ++i) t.v[i] = v[i];                                \
      *reinterpret_cast<Pack16<S, VEC> *>(p) = t;                                                   \
    }                                                                                               \
  };
GGL_INT_VECIO(int32_t, 4)
GGL_INT_VECIO(int64_t, 2)
#undef GGL_INT_VECIO

// RAGGED rows (K not a multiple of the vector width, or a base / stride that is not 16-byte aligned): still one 16-byte
// access per lane.  A row of 47 floats used to take the VEC = 1 kernels — one dword per lane, 64 lanes per row, and a
// wave-wide dword load costs the texture addresser as many cycles as a dwordx4 one — at 3.3 TB/s where K = 48 runs at 5+.
// Here every lane moves 16 bytes from an element-aligned address (global_load_dwordx4 needs dword alignment only; the
// backend emits it for the packed structs below: unaligned access mode), and the row's LAST lane, which would have
// K % VEC elements left, moves back to own the row's last VEC columns instead (`ragged_base`): the columns it shares
// with its neighbour are computed twice, by the same adds in the same order, and stored twice with the same bits.  No
// narrow loads, no shifts, no selects — rounds 2-3 read the tail as up to three dword loads (f32) or one shifted
// 16-byte load with eight 64-bit selects (16-bit rows: 110 registers, 4 wavefronts per SIMD where the aligned kernel
// holds 8).  A ragged launch needs K >= VEC (checked at launch).
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
template <typename S, int VEC, bool RAG> struct RowIO {
  static __device__ __forceinline__ void load(const S *__restrict__ p, S (&v)[VEC], int) { VecIO<S, VEC>::load(p, v); }
  static __device__ __forceinline__ void store(S *__restrict__ p, const S (&v)[VEC], int) { VecIO<S, VEC>::store(p, v); }
};
template <> struct RowIO<float, 4, true> {
  static __device__ __forceinline__ void load(const float *__restrict__ p, float (&v)[4], int) {
    const F4U t = *reinterpret_cast<const F4U *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float *__restrict__ p, const float (&v)[4], int) {
    F4U t;
    t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    *reinterpret_cast<F4U *>(p) = t;
  }
};
template <> struct RowIO<uint16_t, 8, true> {
  static __device__ __forceinline__ void load(const uint16_t *__restrict__ p, uint16_t (&v)[8], int) {
    const H8U t = *reinterpret_cast<const H8U *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = t.v[i];
  }
  static __device__ __forceinline__ void store(uint16_t *__restrict__ p, const uint16_t (&v)[8], int) {
    H8U t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = v[i];
    *reinterpret_cast<H8U *>(p) = t;
  }
};
// first column of the VEC a lane owns when its slot starts at column k0: the row's last VEC columns for the ragged tail
template <int VEC, bool RAG> __device__ __forceinline__ int64_t ragged_base(int64_t K, int64_t k0) {
  return (RAG && K - k0 < VEC) ? K - VEC : k0;
}
template <int VEC, bool RAG> __device__ __forceinline__ int valid_lanes(int64_t K, int64_t kk) {
  return RAG ? (int)((K - kk) < VEC ? (K - kk) : VEC) : VEC;
}

// The argmax witness of a lane's element travels in a 32-bit register: it is an edge position or a source node id, both
// of which index int32 arrays (perm / col), or the "empty" fill (E, or 0), so it always fits; it is widened to the
// reference's int64 at the store.  Eight 64-bit witnesses per lane (the 16-byte 16-bit paths) were 16 registers and two
// selects per element each.
using argreg_t = int32_t;

// ---- reduce sorted positions [beg, end) of `row` for the VEC features starting at kk -------------
template <typename T, int VEC, int OP, int MODE, int IDX, int U, bool RAG = false>
__device__ __forceinline__ void reduce_range(const RPtrs<typename TT<T>::S> &q, const ReduceDims &d,
                                             int64_t row, int64_t beg, int64_t end, int64_t kk,
                                             typename TT<T>::A (&acc)[VEC], argreg_t (&arg)[VEC]) {
  using S = typename TT<T>::S;
  using A = typename TT<T>::A;
  const int64_t K = d.K;
  const int nv = valid_lanes<VEC, RAG>(K, kk);
  const int64_t head = (MODE == MODE_BSPMM) ? kk / d.C : 0;
  // resolve the index mode (compile time unless IDX_RUNTIME)
  const bool seg_perm = seg_like(MODE) && (IDX == IDX_RUNTIME ? q.perm != nullptr : IDX == IDX_PERM);
  const bool has_w = !seg_like(MODE) && (IDX == IDX_RUNTIME ? q.w != nullptr : IDX != IDX_NONE);
  const bool w_perm = has_w && (IDX == IDX_RUNTIME ? (!d.w_by_pos && q.perm != nullptr) : IDX == IDX_PERM);

  auto element = [&](int64_t p, int64_t &xrow, float &wv, int64_t &who) {
    if (seg_like(MODE)) {
      const int64_t e = seg_perm ? (int64_t)q.perm[p] : p;
      xrow = e;
      who = e;
      wv = 1.0f;
    } else {
      const int64_t c = (int64_t)q.col[p];
      xrow = c;
      // the masked max backward looks its bits up by POSITION: its own, or (mask in forward order) the edge's forward
      // position posT[p] — handed down in the aux_rowptr slot, which only the mean backward uses otherwise
      who = MODE == MODE_MAXBWDM ? (q.aux_rowptr ? (int64_t) reinterpret_cast<const int32_t *>(q.aux_rowptr)[p] : p) : c;
      wv = 1.0f;
      if (has_w) {
        const int64_t wi = w_perm ? (int64_t)q.perm[p] : p;
        wv = (MODE == MODE_BSPMM) ? q.w[wi * d.H + head] : q.w[wi];
      }
    }
  };

  auto accumulate = [&](const S (&raw)[VEC], int64_t xrow, float wv, int64_t who) {
    uint32_t mbits = 0;
    if (MODE == MODE_MAXBWDM && !RAG) {
      // a lane's VEC (1 or 4) columns start at a multiple of VEC: their bits sit in ONE word of the edge's mask row
      const uint32_t *mk = reinterpret_cast<const uint32_t *>(q.aux_arg) + who * d.mask_words;
      const int64_t kc = d.mask_col0 + kk;          // (column block of a wider row: the bit index is the full row's)
      mbits = mk[kc >> 5] >> (kc & 31);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      A v = TT<T>::load(raw[i]);
      if (spmm_like(MODE) || MODE == MODE_BSPMM) {
        if (has_w) v = (A)__fmul_rn(wv, (float)v);
      } else if (MODE == MODE_MEANBWD) {
        const int64_t cnt = q.aux_rowptr[xrow + 1] - q.aux_rowptr[xrow];
        v = (A)__fdiv_rn((float)v, (float)cnt);
        if (has_w) v = (A)__fmul_rn((float)v, wv);
      } else if (MODE == MODE_MAXBWDM) {
        if (RAG) {    // (ragged rows: the tail lane's columns may straddle a word)
          const uint32_t *mk = reinterpret_cast<const uint32_t *>(q.aux_arg) + who * d.mask_words;
          const int64_t kc = d.mask_col0 + kk + i;
          if (i < nv && !((mk[kc >> 5] >> (kc & 31)) & 1u)) continue;
        } else if (!((mbits >> i) & 1u)) {
          continue;
        }
        if (has_w) v = (A)__fmul_rn(wv, (float)v);
      } else if (MODE == MODE_MAXBWD || MODE == MODE_MAXBWD32) {
        const int64_t won = MODE == MODE_MAXBWD32 ? (int64_t) reinterpret_cast<const int32_t *>(q.aux_arg)[xrow * K + kk + i]
                                                  : q.aux_arg[xrow * K + kk + i];
        if (won != row) continue;
        if (has_w) v = (A)__fmul_rn(wv, (float)v);
      }
      if (OP == OP_MAX) {
        if (TT<T>::less(acc[i], v)) {
          acc[i] = v;
          arg[i] = (argreg_t)who;
        }
      } else {
        acc[i] = TT<T>::add(acc[i], v);
      }
    }
  };

  int64_t p = beg;
  // (requesting batch i + 1's indices behind batch i's rows — one memory round trip per batch instead of two dependent
  //  ones — was tried in round 4 and buys nothing: products step 75.2-75.4 vs 75.1-75.8 ms, every shape of the op sweep
  //  within noise or slower (max: the extra registers cost a wavefront of occupancy) — the walks are throughput-bound.)
  for (; p + U <= end; p += U) {
    int64_t xrow[U], who[U];
    float wv[U];
    S raw[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) element(p + u, xrow[u], wv[u], who[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) RowIO<S, VEC, RAG>::load(q.x + xrow[u] * d.x_ld + kk, raw[u], nv);
#pragma unroll
    for (int u = 0; u < U; ++u) accumulate(raw[u], xrow[u], wv[u], who[u]);
  }
  // the remaining end - p < U elements: whole batches of 4 first (the narrow kernels walk with U = 16; measured:
  // a 15-way predicated tail costs them 14 %), then ONE partial batch of <= 3 whose loads are issued together
  // and consumed in order (a row of 7 used to pay 1 + 3 dependent round trips — its tail walked one element at
  // a time; with average degrees of 10-50 and sampled blocks of 10 / 25 edges per row most of a row IS tail)
  constexpr int TB = U > 4 ? 4 : U;
  if (U > 4) {
    for (; p + TB <= end; p += TB) {
      int64_t xrow[TB], who[TB];
      float wv[TB];
      S raw[TB][VEC];
#pragma unroll
      for (int u = 0; u < TB; ++u) element(p + u, xrow[u], wv[u], who[u]);
#pragma unroll
      for (int u = 0; u < TB; ++u) RowIO<S, VEC, RAG>::load(q.x + xrow[u] * d.x_ld + kk, raw[u], nv);
#pragma unroll
      for (int u = 0; u < TB; ++u) accumulate(raw[u], xrow[u], wv[u], who[u]);
    }
  }
  const int rem = (int)(end - p);
  if (rem > 0) {
    int64_t xrow[TB], who[TB];
    float wv[TB];
    S raw[TB][VEC];
#pragma unroll
    for (int u = 0; u < TB - 1; ++u)
      if (u < rem) element(p + u, xrow[u], wv[u], who[u]);
#pragma unroll
    for (int u = 0; u < TB - 1; ++u)
      if (u < rem) RowIO<S, VEC, RAG>::load(q.x + xrow[u] * d.x_ld + kk, raw[u], nv);
#pragma unroll
    for (int u = 0; u < TB - 1; ++u)
      if (u < rem) accumulate(raw[u], xrow[u], wv[u], who[u]);
  }
}

template <typename T, int VEC, int OP>
__device__ __forceinline__ void init_acc(typename TT<T>::A (&acc)[VEC], argreg_t (&arg)[VEC],
                                         int64_t arg_fill) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    acc[i] = (OP == OP_MAX) ? TT<T>::lowest() : TT<T>::zero();
    arg[i] = (argreg_t)arg_fill;
  }
}

// accumulate mode: the row starts from what `out` already holds (sum only)
template <typename T, int VEC, int OP, bool RAG = false>
__device__ __forceinline__ void seed_acc(const ReduceDims &d, const typename TT<T>::S *__restrict__ out,
                                         int64_t row, int64_t kk, typename TT<T>::A (&acc)[VEC]) {
  if (OP == OP_SUM && d.accumulate) {
    typename TT<T>::S o[VEC];
    RowIO<typename TT<T>::S, VEC, RAG>::load(out + row * d.out_ld + kk, o, valid_lanes<VEC, RAG>(d.K, kk));
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = TT<T>::load(o[i]);
  }
}

// The epilogue's own operands (bias, the per-row `add` term), requested BEFORE the row is walked: read in
// finish_row they were one more dependent round trip at the end of every row — 7 % of a 13-batch row (measured on
// the 64-column-block launches of the products-sized K = 256 aggregate: 3.65 ms plain, 3.93 ms with a bias).
template <int VEC> struct EpiPre {
  float b[VEC], a[VEC];
  uint32_t rw[4];   // the row's dropout words: ALU work that can run under the walk's memory latency instead of after it
};
template <typename T, int VEC, int MODE, bool RAG>
__device__ __forceinline__ void epi_prefetch(const RPtrs<typename TT<T>::S> &q, const ReduceDims &d, int64_t row,
                                             int64_t kk, EpiPre<VEC> &pre) {
  if (!epi_mode(MODE)) return;
  const int nv = valid_lanes<VEC, RAG>(d.K, kk);
#pragma unroll
  for (int i = 0; i < VEC; ++i) pre.b[i] = (q.epi_bias && (!RAG || i < nv)) ? q.epi_bias[kk + i] : 0.0f;
  if (q.epi_add) RowIO<float, VEC, RAG>::load(q.epi_add + row * d.add_ld + kk, pre.a, nv);
  pre.rw[0] = pre.rw[1] = pre.rw[2] = pre.rw[3] = 0xffffffffu;
  if (d.epi_thresh) {
    const int64_t ev = d.epi_vec;
    const int64_t kg = d.epi_col0 + kk;  // column of acc[0] in the full epi_K-wide row
    const int sh = ev == 4 ? 2 : 0;                       // ev is 1 or 4
    const int64_t KV = (d.epi_K + ev - 1) >> sh;
    if (RAG && VEC > 1 && ev == 1) {  // one Philox word per ELEMENT (epi_K % 4 != 0): component x of its own draw
#pragma unroll
      for (int i = 0; i < (VEC < 4 ? VEC : 4); ++i)
        pre.rw[i] = i < nv ? philox4x32_10((uint64_t)(row * KV + kg + i), (uint64_t)q.epi_rng[1], (uint64_t)q.epi_rng[0]).x
                           : 0xffffffffu;
    } else {
      const U4 u = philox4x32_10((uint64_t)(row * KV + (kg >> sh)), (uint64_t)q.epi_rng[1], (uint64_t)q.epi_rng[0]);
      pre.rw[0] = u.x; pre.rw[1] = u.y; pre.rw[2] = u.z; pre.rw[3] = u.w;
    }
  }
}

// mean / store epilogue of a finished row
template <typename T, int VEC, int OP, int MODE, bool RAG = false>
__device__ __forceinline__ void finish_row(const RPtrs<typename TT<T>::S> &q, const ReduceDims &d,
                                           typename TT<T>::S *__restrict__ out,
                                           int64_t *__restrict__ argout, int64_t K, int64_t row,
                                           int64_t len, int64_t kk, typename TT<T>::A (&acc)[VEC],
                                           const argreg_t (&arg)[VEC], const EpiPre<VEC> &pre) {
  using S = typename TT<T>::S;
  const int nv = valid_lanes<VEC, RAG>(K, kk);
  if (OP == OP_MEAN) {
    if (seg_like(MODE)) {
      // segment_mean_cpu.cpp:67-76: count lives in x's dtype; divide only where count > 1
      const typename TT<T>::A c = TT<T>::count(len);
      if (TT<T>::gt1(c)) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = TT<T>::div(acc[i], c);
      }
    } else {
      // spmm_mean_cpu.cpp:51-58: int64 count cast to float; divide where count > 0
      if (len > 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = (typename TT<T>::A)__fdiv_rn((float)acc[i], (float)len);
      }
    }
  }
  if (epi_mode(MODE)) {
    const uint32_t (&rw)[4] = pre.rw;
    const int64_t ev = d.epi_vec;
    const int64_t kg = d.epi_col0 + kk;  // column of acc[0] in the full epi_K-wide row
    const bool own_words = RAG && VEC > 1 && ev == 1;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = (float)acc[i];
      if (q.epi_add) v = __fadd_rn(v, pre.a[i]);
      if (q.epi_bias && (!RAG || i < nv)) v = __fadd_rn(v, pre.b[i]);
      if (d.epi_relu) v = (v < 0.0f) ? 0.0f : v;
      if (d.epi_thresh) v = (rw[own_words ? i : (int)((kg + i) & (ev - 1))] >= d.epi_thresh) ? __fmul_rn(v, d.epi_scale) : 0.0f;
      acc[i] = (typename TT<T>::A)v;
    }
  }
  S o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = TT<T>::store(acc[i]);
  RowIO<S, VEC, RAG>::store(out + row * d.out_ld + kk, o, nv);
  if (OP == OP_MAX) {
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (!RAG || i < nv) argout[row * K + kk + i] = (int64_t)arg[i];
  }
}

#define GGL_RPTR_PARAMS(S)                                                                         \
  const S *__restrict__ x, const int32_t *__restrict__ perm, const int32_t *__restrict__ col,      \
      const float *__restrict__ w, const int64_t *__restrict__ rowptr,                             \
      const int64_t *__restrict__ aux_rowptr, const int64_t *__restrict__ aux_arg,                 \
      const float *__restrict__ epi_bias, const int64_t *__restrict__ epi_rng,                      \
      const float *__restrict__ epi_add
#define GGL_RPTR_PACK(S) RPtrs<S> q{x, perm, col, w, rowptr, aux_rowptr, aux_arg, epi_bias, epi_rng, epi_add}

// ---- the one launch: chunks of long rows first, then every row with len <= chunk -----------------
// Blocks [0, chunk_blocks) each reduce 4 chunks of long rows (one wavefront per chunk) into the
// partial buffer; they carry the lowest block ids so the hub work is dispatched first and the tail
// of the launch is made of short rows.  Blocks [chunk_blocks, chunk_blocks + nblocks) own rows.
// long_final_kernel then combines the partials of each long row in chunk order.
// Wavefronts per SIMD the register allocator is asked to leave room for (1 = no request).  The 16-bit 8-per-lane maxima
// land a few registers above a step of the occupancy ladder (f16: 76, one wavefront below bf16's 72 — the backend keeps
// two copies of the eight running maxima as live-outs of the divergent walk loop) and these walks are occupancy-bound
// (profiles/r4_negative_results.txt): ask for the step — 72 registers and 24 bytes of scratch per lane outside the walk loop;
// f16 segment_max on the products-sized graph K = 32 / 64 / 128: 4.97 / 5.05 / 7.75 -> 4.59 / 4.84 / 7.41 ms.
template <typename T, int VEC, int OP, int U, bool RAG> constexpr int rr_min_waves() {
  return (sizeof(typename TT<T>::S) == 2 && VEC == 8 && OP == OP_MAX && U == 4 && !RAG) ? 7 : 1;
}
#ifdef GGL_EMULATE
#define GGL_RR_WAVES(T, VEC, OP, U, RAG)
#else
#define GGL_RR_WAVES(T, VEC, OP, U, RAG) __attribute__((amdgpu_waves_per_eu(rr_min_waves<T, VEC, OP, U, RAG>(), 8)))
#endif
template <typename T, int VEC, int OP, int MODE, int IDX, bool UNIFORM, int U, bool RAG = false>
__global__ __launch_bounds__(kBlock) GGL_RR_WAVES(T, VEC, OP, U, RAG) void row_reduce_kernel(GGL_RPTR_PARAMS(typename TT<T>::S),
                                                            const int32_t *__restrict__ row_order,
                                                            const int32_t *__restrict__ long_rows,
                                                            const int64_t *__restrict__ chunk_ptr,
                                                            typename TT<T>::S *__restrict__ partial,
                                                            int64_t *__restrict__ partial_arg,
                                                            typename TT<T>::S *__restrict__ out,
                                                            int64_t *__restrict__ argout,
                                                            const ReduceDims d) {
  using S = typename TT<T>::S;
  using A = typename TT<T>::A;
  GGL_RPTR_PACK(S);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  if (block_id() < d.chunk_blocks) {
    const int c32 = __builtin_amdgcn_readfirstlane((int)(block_id() * kWavesPerBlock + wave));
    const int64_t cid = (int64_t)(uint32_t)c32;
    if (cid >= d.n_chunks) return;
    // owning long row: last j with chunk_ptr[j] <= cid
    int64_t lo = 0, hi = d.n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= cid) lo = mid; else hi = mid - 1;
    }
    const int64_t row = long_rows[lo];
    const int64_t local = cid - chunk_ptr[lo];
    const int64_t rbeg = rowptr[row], rend = rowptr[row + 1];
    const int64_t beg = rbeg + local * d.chunk;
    const int64_t end = (beg + d.chunk < rend) ? beg + d.chunk : rend;
    for (int64_t k0 = (int64_t)lane * VEC; k0 < d.K; k0 += (int64_t)kWave * VEC) {
      const int64_t kk = ragged_base<VEC, RAG>(d.K, k0);
      A acc[VEC];
      argreg_t arg[VEC];
      init_acc<T, VEC, OP>(acc, arg, d.arg_fill);
      reduce_range<T, VEC, OP, MODE, IDX, U, RAG>(q, d, row, beg, end, kk, acc, arg);
      S o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = TT<T>::store(acc[i]);
      const int nv = valid_lanes<VEC, RAG>(d.K, kk);
      RowIO<S, VEC, RAG>::store(partial + cid * d.K + kk, o, nv);
      if (OP == OP_MAX) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (!RAG || i < nv) partial_arg[cid * d.K + kk + i] = (int64_t)arg[i];
      }
    }
    return;
  }
  if (block_id() - d.chunk_blocks >= d.nblocks) return;  // padding blocks of a folded (2-D) grid
  const int64_t blk = xcd_remap(block_id() - d.chunk_blocks, d.nblocks, d.swizzle);
  const int L = 1 << d.logL;
  int64_t slot;
  int li;
  if (UNIFORM) {  // one wavefront per row: everything about the row is wave-uniform (scalar path)
    const int r32 = __builtin_amdgcn_readfirstlane((int)(blk * kWavesPerBlock + wave));
    slot = (int64_t)(uint32_t)r32;  // nblocks * 4 < 2^32 is checked at launch
    li = lane;
  } else {
    const int rows_per_wave = kWave >> d.logL;
    slot = (blk * kWavesPerBlock + wave) * rows_per_wave + (lane >> d.logL);
    li = lane & (L - 1);
  }
  if (slot >= d.N) return;
  // row_order (rows sorted by length, longest first) packs rows of similar length into one
  // wavefront when several rows share it, and starts the heavy rows early
  const int64_t row = row_order ? (int64_t)row_order[slot] : slot;
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  const int64_t len = end - beg;
  if (len > d.chunk) return;  // long row: reduced by the chunk blocks above + long_final_kernel
  for (int64_t k0 = (int64_t)li * VEC; k0 < d.K; k0 += (int64_t)L * VEC) {
    const int64_t kk = ragged_base<VEC, RAG>(d.K, k0);
    A acc[VEC];
    argreg_t arg[VEC];
    init_acc<T, VEC, OP>(acc, arg, d.arg_fill);
    seed_acc<T, VEC, OP, RAG>(d, out, row, kk, acc);
    EpiPre<VEC> pre;
    epi_prefetch<T, VEC, MODE, RAG>(q, d, row, kk, pre);
    reduce_range<T, VEC, OP, MODE, IDX, U, RAG>(q, d, row, beg, end, kk, acc, arg);
    finish_row<T, VEC, OP, MODE, RAG>(q, d, out, argout, d.K, row, len, kk, acc, arg, pre);
  }
}

template <typename T, int OP, int MODE>
__global__ __launch_bounds__(kBlock) void long_final_kernel(const int64_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ long_rows,
                                                            const int64_t *__restrict__ chunk_ptr,
                                                            const typename TT<T>::S *__restrict__ partial,
                                                            const int64_t *__restrict__ partial_arg,
                                                            typename TT<T>::S *__restrict__ out,
                                                            int64_t *__restrict__ argout,
                                                            const float *__restrict__ epi_bias,
                                                            const int64_t *__restrict__ epi_rng,
                                                            const float *__restrict__ epi_add,
                                                            const ReduceDims d) {
  using A = typename TT<T>::A;
  RPtrs<typename TT<T>::S> q{};
  q.epi_bias = epi_bias;
  q.epi_rng = epi_rng;
  q.epi_add = epi_add;
  const int64_t j = block_id();  // one block per long row
  if (j >= d.n_long) return;
  const int64_t row = long_rows[j];
  const int64_t c0 = d.exact_long ? j : chunk_ptr[j], c1 = d.exact_long ? j + 1 : chunk_ptr[j + 1];
  const int64_t len = rowptr[row + 1] - rowptr[row];
  for (int64_t k = threadIdx.x; k < d.K; k += kBlock) {
    A acc[1];
    argreg_t arg[1];
    init_acc<T, 1, OP>(acc, arg, d.arg_fill);
    seed_acc<T, 1, OP>(d, out, row, k, acc);
    for (int64_t c = c0; c < c1; ++c) {
      const A v = TT<T>::load(partial[c * d.K + k]);
      if (OP == OP_MAX) {
        if (TT<T>::less(acc[0], v)) {  // strict <: the earliest chunk (smallest e) keeps ties
          acc[0] = v;
          arg[0] = (argreg_t)partial_arg[c * d.K + k];
        }
      } else {
        acc[0] = TT<T>::add(acc[0], v);
      }
    }
    EpiPre<1> pre;
    epi_prefetch<T, 1, MODE, false>(q, d, row, k, pre);
    finish_row<T, 1, OP, MODE>(q, d, out, argout, d.K, row, len, k, acc, arg, pre);
  }
}

// ---- host-side dispatch -------------------------------------------------------------------------
struct ReduceArgs {  // host-side bundle: everything one logical op needs
  const void *x;
  const int32_t *perm;
  const int32_t *col;
  const float *w;
  int w_by_pos;
  const int64_t *rowptr;
  int64_t N, K, E;
  void *out;
  int64_t *arg;
  int64_t arg_fill;
  int64_t chunk;
  int64_t H, C;
  const int64_t *aux_rowptr;
  const int64_t *aux_arg;
  const int32_t *long_rows;
  const int64_t *chunk_ptr;
  const int32_t *row_order;
  const int32_t *long_order;
  int64_t max_len;
  int64_t xcd_run_rows;
  int64_t n_long, n_chunks;
  void *partial;
  int64_t *partial_arg;
  int64_t x_ld, out_ld;    // 0 = dense (K)
  int accumulate;
  const float *epi_bias;   // epilogue modes
  const int64_t *epi_rng;
  int epi_relu;
  uint32_t epi_thresh;
  float epi_scale;
  const float *epi_add;
  int64_t add_ld;
  int64_t epi_K, epi_col0; // 0 / 0 = the launch covers whole rows
  // column-block launches whose hub rows are walked ONCE per aggregate (launch_f32_cols): 0 = the whole op (hub walk, row
  // walk, long_final); 1 = the row walk only (the hub walk over the full width is already in flight, long_final comes
  // later); 2 = join the hub walk (if `hub_forked`) + long_final only
  int phase;
  int hub_forked;
  int64_t mask_words, mask_col0;   // MODE_MAXBWDM
};

static inline int pow2_ceil_log2(int64_t v) {
  int l = 0;
  while (((int64_t)1 << l) < v) ++l;
  return l;
}

#define GGL_RPTR_ARGS(S)                                                                           \
  static_cast<const S *>(a.x), a.perm, a.col, a.w, a.rowptr, a.aux_rowptr, a.aux_arg, a.epi_bias, a.epi_rng, a.epi_add

// f32 sums whose long rows are reduced in the reference's serial order instead of chunk by chunk (hubf32.hip): every
// mode whose value is "an element, times its weight" — segment sum / mean, SpMM sum / mean, bspmm, with or without the
// epilogue.  (max has no rounding; the mean / max backward walks keep their chunks.)
template <typename T, int OP, int MODE> constexpr bool exact_long_mode() {
  return OP != OP_MAX && ((std::is_same<T, float>::value && (seg_like(MODE) || spmm_like(MODE) || MODE == MODE_BSPMM)) ||
                          (std::is_same<T, double>::value && (MODE == MODE_SEG)));
}

// ... and whether this launch takes it (positions travel as int32 in the hub kernel's registers, like the plan's own perm
// entries; one add chain of 10^7 elements: no)
template <typename T, int OP, int MODE> static bool exact_long_applies(const ReduceArgs &a) {
#ifdef GGL_EMULATE
  return false;
#else
  if constexpr (exact_long_mode<T, OP, MODE>())
    return a.n_long > 0 && options().exact_long_rows != 0 && a.E < ((int64_t)1 << 31) &&
           (options().exact_long_max <= 0 || a.max_len <= options().exact_long_max);
  return false;
#endif
}

#ifndef GGL_EMULATE
template <typename T, int MODE> static HubF32Args hub_args_of(const ReduceArgs &a, int64_t x_ld) {
  HubF32Args h{};
  constexpr int kWords = std::is_same<T, double>::value ? 2 : 1;   // doubles travel as pairs of 4-byte words (hubf32.hip)
  h.f64 = kWords == 2 ? 1 : 0;
  h.x = reinterpret_cast<const float *>(a.x);
  h.x_ld = x_ld * kWords;
  h.perm = a.perm;
  h.col = seg_like(MODE) ? nullptr : a.col;
  h.w = seg_like(MODE) ? nullptr : a.w;
  h.w_by_pos = a.w_by_pos;
  h.H = a.H;
  h.C = (MODE == MODE_BSPMM) ? a.C : 0;
  h.rowptr = a.rowptr;
  h.long_rows = a.long_rows;
  h.long_order = a.long_order;
  h.n_long = a.n_long;
  h.K = a.K * kWords;
  h.partial = static_cast<float *>(a.partial);
  h.avg_long_len = a.n_chunks * a.chunk / (a.n_long > 0 ? a.n_long : 1);   // (chunks are full but the last of a row)
  return h;
}
#endif

template <typename T, int VEC, int OP, int MODE, int IDX, bool RAG = false>
static int launch_idx(const ReduceArgs &a_in, ReduceDims d, hipStream_t stream) {
  using S = typename TT<T>::S;
  ReduceArgs a = a_in;
  const bool uniform = !RAG && (d.logL == 6) && std::is_same<T, float>::value;   // (ragged rows: K <= 128 only)
  S *out = static_cast<S *>(a.out);
  int forked = a.phase == 2 ? a.hub_forked : 0;     // the hub launch's join token (0 = nothing to join)
  const bool exact = exact_long_applies<T, OP, MODE>(a);
#ifdef GGL_EMULATE
  // the host build walks a row with one thread anyway: no chunks at all IS the reference's serial order — for every
  // summing mode and dtype (f64 and the backward walks included), not only the ones the GPU's hub kernel covers
  if (OP != OP_MAX && a.n_long > 0 && options().exact_long_rows != 0) {
    a.n_long = 0; a.n_chunks = 0;
    d.n_long = 0; d.n_chunks = 0;
    d.chunk = (int64_t)1 << 62;
  }
#endif
  GGL_REQUIRE(a.phase == 0 || exact, GGL_EINVAL, "phased launches are for the exact hub walk only");
  if (a.n_long > 0)
    GGL_REQUIRE(a.partial != nullptr, GGL_EWORKSPACE, "plan has long rows but no partial buffer");
  d.chunk_blocks = (a.n_long > 0 && !exact) ? ceil_div(a.n_chunks, kWavesPerBlock) : 0;
  d.exact_long = exact ? 1 : 0;
#ifndef GGL_EMULATE
  if constexpr (exact_long_mode<T, OP, MODE>()) {
    if (exact && a.phase == 0) {
      const HubF32Args h = hub_args_of<T, MODE>(a, d.x_ld);
      const int rc = hub_f32_launch(h, stream, options().exact_side_stream != 0, &forked);
      if (rc) return rc;
    }
  }
#endif
  const int64_t grid = d.chunk_blocks + d.nblocks;
  GGL_REQUIRE(grid < ((int64_t)1 << 31), GGL_EINVAL, "grid too large");
  const int32_t *order = (options().row_order && (!uniform || options().row_order > 1)) ? a.row_order : nullptr;
#define GGL_RR_ARGS GGL_RPTR_ARGS(S), order, a.long_rows, a.chunk_ptr, static_cast<S *>(a.partial), a.partial_arg, out, a.arg, d
  if (a.phase == 2) {
    // (join + long_final only)
  } else if (uniform) {
    if constexpr (!RAG) {
      // the f32 wave-per-row kernels; U = 8 only for the dominant SpMM-sum (A/B knob)
      if (std::is_same<T, float>::value && VEC == 4 && OP == OP_SUM && spmm_like(MODE) &&
          options().unroll >= 8) {
        GGL_LAUNCH((row_reduce_kernel<T, VEC, OP, MODE, IDX, std::is_same<T, float>::value, 8>), grid,
                   kBlock, stream, GGL_RR_ARGS);
      } else {
        GGL_LAUNCH((row_reduce_kernel<T, VEC, OP, MODE, IDX, std::is_same<T, float>::value, 4>), grid,
                   kBlock, stream, GGL_RR_ARGS);
      }
    }
  } else if (!RAG && (OP != OP_MAX || options().unroll_narrow_max) && d.logL <= 2 && options().unroll_narrow > 4) {
    // narrow rows (<= 4 lanes per row, K <= 16 floats): a lane moves 16 bytes per element, so the walk is
    // latency-bound; 16 elements in flight per lane instead of 4 (Reddit-sized segment_sum: K = 1
    // 1.73 -> 1.23 ms, K = 8 2.56 -> 2.12 ms, profiles/r1_smallk_probe.txt).  Not for max: its int64
    // argmax registers make the deeper unroll slower (2.79 -> 3.27 ms).
    GGL_LAUNCH((row_reduce_kernel<T, VEC, OP, MODE, IDX, false, 16>), grid, kBlock, stream, GGL_RR_ARGS);
  } else {
    GGL_LAUNCH((row_reduce_kernel<T, VEC, OP, MODE, IDX, false, 4, RAG>), grid, kBlock, stream, GGL_RR_ARGS);
  }
#undef GGL_RR_ARGS
  GGL_LAUNCH_CHECK();
  if (a.phase == 1) return GGL_OK;
#ifndef GGL_EMULATE
  if (forked) {
    const int rc = hub_f32_join(stream, forked);
    if (rc) return rc;
  }
#endif
  if (a.n_long > 0) {
    GGL_LAUNCH((long_final_kernel<T, OP, MODE>), a.n_long, kBlock, stream, a.rowptr, a.long_rows,
               a.chunk_ptr, static_cast<const S *>(a.partial), (const int64_t *)a.partial_arg, out,
               a.arg, a.epi_bias, a.epi_rng, a.epi_add, d);
    GGL_LAUNCH_CHECK();
  }
  return GGL_OK;
}

// STATIC_IDX: compile-time index modes (hot f32 segment / SpMM kernels); otherwise IDX_RUNTIME
template <typename T, int VEC, int OP, int MODE, bool STATIC_IDX, bool RAG = false>
static int launch_typed(const ReduceArgs &a, hipStream_t stream) {
  ReduceDims d{};
  d.N = a.N; d.K = a.K; d.E = a.E; d.arg_fill = a.arg_fill; d.chunk = a.chunk; d.H = a.H; d.C = a.C;
  d.n_long = a.n_long; d.n_chunks = a.n_chunks; d.w_by_pos = a.w_by_pos;
  d.x_ld = a.x_ld > 0 ? a.x_ld : a.K; d.out_ld = a.out_ld > 0 ? a.out_ld : a.K; d.accumulate = a.accumulate;
  d.epi_relu = a.epi_relu; d.epi_thresh = a.epi_thresh; d.epi_scale = a.epi_scale;
  d.epi_K = a.epi_K > 0 ? a.epi_K : a.K; d.epi_col0 = a.epi_col0;
  d.epi_vec = (d.epi_K % 4 == 0) ? 4 : 1;
  d.add_ld = a.add_ld > 0 ? a.add_ld : a.K;
  d.mask_words = a.mask_words;
  d.mask_col0 = a.mask_col0;
  const int64_t kv = ceil_div(a.K, VEC);
  d.logL = pow2_ceil_log2(kv < kWave ? kv : kWave);
  if (d.logL > 6) d.logL = 6;
  const int rows_per_block = kWavesPerBlock * (kWave >> d.logL);
  d.nblocks = ceil_div(a.N, rows_per_block);
  d.swizzle = (int)options().xcd_swizzle;
  if (a.xcd_run_rows > 0 && !d.swizzle) {   // a locality-ordered graph: runs of consecutive row blocks per XCD
    const int64_t run_blocks = a.xcd_run_rows / rows_per_block;
    if (run_blocks >= 2 && d.nblocks >= 16 * run_blocks) d.swizzle = (int)run_blocks;
  }
  GGL_REQUIRE(d.nblocks < ((int64_t)1 << 30), GGL_EINVAL, "too many rows for one launch");
  GGL_REQUIRE(!RAG || (a.K >= VEC && !a.accumulate), GGL_EINVAL, "ragged rows: K >= the lane vector, no accumulate");
  if (a.N <= 0 || a.K <= 0) return GGL_OK;
  if constexpr (!STATIC_IDX || RAG) {   // (the ragged kernels resolve the index mode at run time: one variant each)
    return launch_idx<T, VEC, OP, MODE, IDX_RUNTIME, RAG>(a, d, stream);
  } else if constexpr (seg_like(MODE)) {
    if (a.perm) return launch_idx<T, VEC, OP, MODE, IDX_PERM>(a, d, stream);
    return launch_idx<T, VEC, OP, MODE, IDX_DIRECT>(a, d, stream);
  } else {
    if (!a.w) return launch_idx<T, VEC, OP, MODE, IDX_NONE>(a, d, stream);
    if (a.w_by_pos || !a.perm) return launch_idx<T, VEC, OP, MODE, IDX_DIRECT>(a, d, stream);
    return launch_idx<T, VEC, OP, MODE, IDX_PERM>(a, d, stream);
  }
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int OP, int MODE>
static int launch_f32(const ReduceArgs &a, hipStream_t stream) {
  constexpr bool kStatic = (seg_like(MODE) || spmm_like(MODE) || MODE == MODE_MAXBWDM);
  const bool vec4 = !options().force_generic && (a.K % 4 == 0) && aligned16(a.x) &&
                    aligned16(a.out) && (!a.partial || aligned16(a.partial)) &&
                    (a.x_ld % 4 == 0) && (a.out_ld % 4 == 0) && (MODE != MODE_BSPMM || a.C % 4 == 0) &&
                    (!a.epi_add || (aligned16(a.epi_add) && a.add_ld % 4 == 0));
  if (vec4) return launch_typed<float, 4, OP, MODE, kStatic>(a, stream);
  // rows that are not made of aligned float4s: four floats per lane all the same, ragged last lane (see RowIO).
  // Measured on the products-sized graph (profiles/r2_ragged_rows.txt; same bits): gspmm max K = 101 12.5 -> 8.8 ms,
  // K = 41 6.2 -> 5.5 ms, segment_sum [E, 47] 7.46 -> 7.22 ms; below 32 columns the VEC = 1 kernels with 16 loads in
  // flight win by 3x and wave-per-row widths (K > 128) lose 7 %, so only 32 <= K <= 128 takes it.
  // (segment_max too since the tail lane owns the row's last four columns instead of 1-3 narrow loads: [E, 47] 8.16 -> 7.64 ms;
  //  with the narrow tail and 64-bit witnesses it lost, 7.9 -> 9.1)
  if constexpr (seg_like(MODE) || spmm_like(MODE)) {
    // (not onto an existing result: the tail lane's shared columns would be seeded from `out` after its neighbour has
    //  stored them wherever lanes do not run in lockstep — the host build — or a row takes more than one pass)
    if (!options().force_generic && options().ragged4 && a.K >= 32 && a.K <= 128 && !a.accumulate &&
        (OP != OP_MAX || spmm_like(MODE) || options().ragged_max))
      return launch_typed<float, 4, OP, MODE, kStatic, true>(a, stream);
  }
  return launch_typed<float, 1, OP, MODE, kStatic>(a, stream);
}

// Wide f32 aggregates run as launches over 64-column blocks of the same matrices (row strides passed down: no copy).
// A 256-byte slice keeps four times as many distinct hub rows in the 4 MiB L2 of an XCD as a 1 KiB row does, and L2
// hits are the only bytes of this walk that do not cross the fabric (profiles/r2_mall_probe.txt): products-sized
// graph, K = 256: 16.5 -> 14.8 ms, K = 128: 8.0 -> 7.5 ms; 32-wide blocks lose again (17.3 ms: a 128-byte line per
// gather whatever the width), as do blocks that are not line multiples (K = 96 as 2 x 48: 5.5 -> 7.3 ms).  The
// columns of a row are independent sums: same bits; the dropout word of element (row, col) is the full-width one
// (epi_K / epi_col0).  Small graphs (an arxiv-sized launch is 0.3 ms) take two 128-wide blocks at most.
// `xcd_run_rows` > 0 (the plan's node order carries locality: every XCD walks runs of consecutive rows): the L2 hits then come
// from the communities' own rows, not from keeping hub rows resident, and narrower blocks only re-read the indices — blocks twice
// as wide (products-sized planted-community graph in its native order, K = 256, profiles/r6_planted_knobs.txt: 32 / 64 / 128 /
// 256-wide blocks 11.8 / 9.8 / 9.3 / 9.7 ms; the randomly labelled R-MAT graph keeps 64: 13.4 vs 14.7 one launch).
static int64_t col_block_width(int64_t E, int64_t K, int64_t N, int64_t xcd_run_rows = 0) {   // 0 = one launch
  // the blocks pay through L2 reuse of hub rows: a walk with few edges per row (a rank's local-source block of the
  // papers100M-sized partition: 4.6) has little to reuse and only re-reads its indices per block (10.9 -> 12.6 ms)
  if (N > 0 && E < options().col_block_min_degree * N) return 0;
  int64_t bw = options().col_block;
  if (bw > 0 && E < options().col_block_min_edges) bw *= 2;
  else if (bw > 0 && xcd_run_rows > 0 && K >= 4 * bw && K % (2 * bw) == 0) bw *= 2;
  if (bw <= 0 || bw % 4 != 0 || K < 2 * bw || K % 4 != 0) return 0;
  // up to 256 columns a row is walked once by its lane group: blocks must tile it exactly (a narrow remainder block
  // would cost a launch of its own for a few columns).  Wider rows are walked once per 256 columns anyway, the last
  // pass with most lanes idle (K = 604: 55 ms where 19 lines per edge should take 35): full blocks + a remainder block
  if (K % bw != 0 && K <= 256) return 0;
  return bw;
}
// launches one f32 SpMM-sum / mean over E edges and K columns is made of (bench.py's roofline leg reports per launch)
extern "C" int64_t ggl_spmm_col_blocks(int64_t E, int64_t K, int64_t N) {
  const int64_t bw = col_block_width(E, K, N);
  return bw > 0 ? (K + bw - 1) / bw : 1;
}

extern "C" int64_t ggl_spmm_col_blocks_plan(const ggl_segplan_t *plan, int64_t K) {
  if (plan == nullptr) return 1;
  const int64_t bw = col_block_width(plan->E, K, plan->N, plan->xcd_run_rows);
  return bw > 0 ? (K + bw - 1) / bw : 1;
}

template <int OP, int MODE>
static int launch_f32_cols(const ReduceArgs &a0, hipStream_t stream) {
  static_assert(OP != OP_MAX && (MODE == MODE_SPMM || MODE == MODE_SPMM_EPI || MODE == MODE_MAXBWDM),
                "column blocks: sum / mean SpMM and the masked max backward (a sum) only");
  const int64_t bw = col_block_width(a0.E, a0.K, a0.N, a0.xcd_run_rows);
  if (bw <= 0 || a0.N <= 0) return launch_f32<OP, MODE>(a0, stream);
  // The hub rows are walked ONCE per aggregate, over the full width (round 5): one hub launch forked in front of the first
  // column block, every slab of every hub row an independent workgroup — the K / 64 add chains of the longest row run
  // side by side instead of one per column-block launch, each of which used to end 0.2-0.5 ms after its row walk — joined
  // before ONE long_final over the full width.  The partial buffer holds n_chunks >= n_long full-width rows.
  bool one_hub = false;
  int forked = 0;
#ifndef GGL_EMULATE
  // hub_one_launch: 1 = always, 0 = never (a hub launch per block), 2 = where the plan says its long rows lead the id range
  // (xcd_run_rows < 0: a degree-sorted node order).  Measured, K = 256, products-sized graph (r5_hub_alone.txt): random order
  // 13.6 ms per block-wise aggregate vs 13.8-14.0 in one launch; degree order 14.3 vs 14.1.
  const int64_t ohl = options().hub_one_launch;
  if ((ohl == 1 || (ohl == 2 && a0.xcd_run_rows < 0)) && exact_long_applies<float, OP, MODE>(a0)) {
    GGL_REQUIRE(a0.partial != nullptr, GGL_EWORKSPACE, "plan has long rows but no partial buffer");
    const HubF32Args h = hub_args_of<float, MODE>(a0, a0.x_ld > 0 ? a0.x_ld : a0.K);
    const int rc = hub_f32_launch(h, stream, options().exact_side_stream != 0, &forked);
    if (rc) return rc;
    one_hub = true;
  }
#endif
  for (int64_t c0 = 0; c0 < a0.K; c0 += bw) {
    ReduceArgs a = a0;
    a.phase = one_hub ? 1 : 0;
    a.K = (a0.K - c0) < bw ? (a0.K - c0) : bw;
    a.x = static_cast<const float *>(a0.x) + c0;
    a.out = static_cast<float *>(a0.out) + c0;
    a.x_ld = a0.x_ld > 0 ? a0.x_ld : a0.K;
    a.out_ld = a0.out_ld > 0 ? a0.out_ld : a0.K;
    if (a0.epi_bias) a.epi_bias = a0.epi_bias + c0;
    if (a0.epi_add) {
      a.epi_add = a0.epi_add + c0;
      a.add_ld = a0.add_ld > 0 ? a0.add_ld : a0.K;
    }
    a.epi_K = a0.epi_K > 0 ? a0.epi_K : a0.K;
    a.epi_col0 = a0.epi_col0 + c0;
    a.mask_col0 = a0.mask_col0 + c0;
    const int rc = launch_f32<OP, MODE>(a, stream);
    if (rc) return rc;
  }
  if (one_hub) {
    ReduceArgs a = a0;
    a.phase = 2;
    a.hub_forked = forked;
    return launch_f32<OP, MODE>(a, stream);
  }
  return GGL_OK;
}

// bspmm over the same 64-column blocks where a block lies INSIDE one head (the block width divides C): the kernels compute a
// column's head as k / C with k counted from the block's first column, so the weight pointer moves to the block's head and
// everything else is the strided launch above.  Products-sized graph, forward: 1 x 256 16.0 -> 14.8 ms.  Blocks of several
// narrow heads were measured too and LOSE (16 x 16: 17.4 -> 18.2 ms, 32 x 8: 19.1 -> 22.7 — every block launch fetches the
// edge's whole [H] weight row for the few heads it uses): those shapes stay one launch.
static int launch_bspmm_cols(const ReduceArgs &a0, hipStream_t stream) {
  int64_t bw = col_block_width(a0.E, a0.K, a0.N);
  if (bw > 0 && a0.C % bw != 0) bw = 0;
  if (bw <= 0 || a0.N <= 0 || !a0.w) return launch_f32<OP_SUM, MODE_BSPMM>(a0, stream);
  for (int64_t c0 = 0; c0 < a0.K; c0 += bw) {
    ReduceArgs a = a0;
    a.K = (a0.K - c0) < bw ? (a0.K - c0) : bw;
    a.x = static_cast<const float *>(a0.x) + c0;
    a.out = static_cast<float *>(a0.out) + c0;
    a.x_ld = a0.x_ld > 0 ? a0.x_ld : a0.K;
    a.out_ld = a0.out_ld > 0 ? a0.out_ld : a0.K;
    a.w = a0.w + c0 / a0.C;          // [., H] rows: the block's first head (H stays the row stride)
    const int rc = launch_f32<OP_SUM, MODE_BSPMM>(a, stream);
    if (rc) return rc;
  }
  return GGL_OK;
}

// 16-byte vector path usable: K a multiple of the vector width and every base pointer 16-byte aligned
static bool wide_ok(const ReduceArgs &a, int vec) {
  return !options().force_generic && a.K % vec == 0 && aligned16(a.x) && aligned16(a.out) &&
         (!a.partial || aligned16(a.partial)) && a.x_ld % vec == 0 && a.out_ld % vec == 0;
}

// 16-bit rows that are not made of aligned 16-byte pieces: eight elements per lane all the same (RowIO<uint16_t, 8, true>),
// from 12 columns up (below that one lane per element with 16 loads in flight wins)
template <int OP> static bool ragged16_ok(const ReduceArgs &a) {
  // (max: above 16 columns, like the aligned rows — since the tail lane owns the row's last eight columns (58 registers
  //  instead of 110) [E, 47] f16 max 9.51 -> 6.38 ms, bf16 9.38 -> 6.20, [E, 100] f16 12.8 -> 8.6; sums: f16 6.03 -> 5.16,
  //  bf16 [E, 100] 10.3 -> 8.8.  With the shifted-load tail it took 72 columns for the maxima to win.)
  return (OP != OP_MAX || a.K >= 72 || (options().ragged_max && a.K > 16)) && !options().force_generic && options().ragged4 &&
         a.K >= 12 &&
         !a.accumulate;
}

template <int OP> static bool narrow16(const ReduceArgs &a) { return a.K <= 8 || (OP == OP_MAX && a.K <= 16); }

template <int OP>
static int launch_seg(int dtype, const ReduceArgs &a, hipStream_t stream) {
  switch (dtype) {
    case GGL_F32: return launch_f32<OP, MODE_SEG>(a, stream);
    case GGL_F64:
      if (wide_ok(a, 2)) return launch_typed<double, 2, OP, MODE_SEG, false>(a, stream);
      return launch_typed<double, 1, OP, MODE_SEG, false>(a, stream);
    // 16-bit rows of <= 8 columns (max: <= 16) walk with ONE element per lane: eight per lane would leave 1-2 lanes per
    // row, i.e. 32-64 rows of very different lengths serialised inside one wavefront (products-sized graph, f16:
    // K = 8 sum 4.64 -> 2.93 ms, max 5.90 -> 3.01; K = 16 max 5.27 -> 3.81; from K = 24 up the 16-byte lanes win —
    // profiles/r4_narrow16_probe.txt).  Same summation order either way: same bits.
    case GGL_F16:
      if (wide_ok(a, 8) && !narrow16<OP>(a)) return launch_typed<f16_t, 8, OP, MODE_SEG, false>(a, stream);
      if (ragged16_ok<OP>(a)) return launch_typed<f16_t, 8, OP, MODE_SEG, false, true>(a, stream);
      return launch_typed<f16_t, 1, OP, MODE_SEG, false>(a, stream);
    case GGL_BF16:
      if (wide_ok(a, 8) && !narrow16<OP>(a)) return launch_typed<bf16_t, 8, OP, MODE_SEG, false>(a, stream);
      if (ragged16_ok<OP>(a)) return launch_typed<bf16_t, 8, OP, MODE_SEG, false, true>(a, stream);
      return launch_typed<bf16_t, 1, OP, MODE_SEG, false>(a, stream);
    case GGL_U8: return launch_typed<uint8_t, 1, OP, MODE_SEG, false>(a, stream);
    case GGL_I8: return launch_typed<int8_t, 1, OP, MODE_SEG, false>(a, stream);
    case GGL_I16: return launch_typed<int16_t, 1, OP, MODE_SEG, false>(a, stream);
    case GGL_I32:
      if (wide_ok(a, 4) && a.K >= 16) return launch_typed<int32_t, 4, OP, MODE_SEG, false>(a, stream);
      return launch_typed<int32_t, 1, OP, MODE_SEG, false>(a, stream);
    case GGL_I64:
      if (wide_ok(a, 2) && a.K >= 8) return launch_typed<int64_t, 2, OP, MODE_SEG, false>(a, stream);
      return launch_typed<int64_t, 1, OP, MODE_SEG, false>(a, stream);
    default: set_error("unsupported dtype code %d", dtype); return GGL_EDTYPE;
  }
}

static int fill_plan(ReduceArgs &a, const ggl_segplan_t *plan, int dtype, int64_t K, bool with_arg) {
  GGL_REQUIRE(plan != nullptr && plan->rowptr != nullptr, GGL_EINVAL, "plan / rowptr is NULL");
  GGL_REQUIRE(plan->chunk > 0, GGL_EINVAL, "plan->chunk must be > 0");
  GGL_REQUIRE(K >= 0 && plan->N >= 0 && plan->E >= 0, GGL_EINVAL, "negative size");
  a.rowptr = plan->rowptr;
  a.perm = plan->perm;
  a.N = plan->N;
  a.E = plan->E;
  a.K = K;
  a.chunk = plan->chunk;
  a.long_rows = plan->long_rows;
  a.chunk_ptr = plan->chunk_ptr;
  a.row_order = plan->row_order;
  a.long_order = plan->long_order;
  a.max_len = plan->max_len;
  a.xcd_run_rows = plan->xcd_run_rows;
  a.n_long = plan->n_long;
  a.n_chunks = plan->n_chunks;
  a.partial = plan->partial;
  a.partial_arg = nullptr;
  if (plan->n_long > 0) {
    GGL_REQUIRE(plan->long_rows && plan->chunk_ptr && plan->partial, GGL_EWORKSPACE,
                "plan has long rows but long_rows/chunk_ptr/partial is NULL");
    if (with_arg) {
      size_t off = (size_t)plan->n_chunks * (size_t)K * dtype_size(dtype);
      off = (off + 15) & ~(size_t)15;
      a.partial_arg = reinterpret_cast<int64_t *>(static_cast<char *>(plan->partial) + off);
    }
  }
  return GGL_OK;
}

}  // namespace ggl

using namespace ggl;

extern "C" size_t ggl_partial_bytes(int dtype, int64_t n_chunks, int64_t K, int with_arg) {
  if (n_chunks <= 0 || K <= 0) return 0;
  size_t b = (size_t)n_chunks * (size_t)K * dtype_size(dtype);
  b = (b + 15) & ~(size_t)15;
  if (with_arg) b += (size_t)n_chunks * (size_t)K * 8;
  return b;
}

// ---- segment ops ---------------------------------------------------------------------------------
extern "C" int ggl_segment_sum(int dtype, const void *x, const ggl_segplan_t *plan, int64_t K,
                               void *out, void *stream) {
  ReduceArgs a{};
  int rc = fill_plan(a, plan, dtype, K, false);
  if (rc) return rc;
  GGL_REQUIRE((x || plan->E * K == 0) && (out || plan->N * K == 0), GGL_EINVAL, "x/out is NULL");
  a.x = x;
  a.out = out;
  return launch_seg<OP_SUM>(dtype, a, as_stream(stream));
}

// Strided / accumulating form: x rows x_ld elements apart, out rows out_ld apart (both >= K; 0 = K), and
// with accumulate != 0 the sums are added to what out holds (the row's previous value comes first in the
// summation order).  Used for column blocks of a wider matrix and for adding a second edge set in place.
extern "C" int ggl_segment_sum_ex(int dtype, const void *x, int64_t x_ld, const ggl_segplan_t *plan,
                                  int64_t K, void *out, int64_t out_ld, int accumulate, void *stream) {
  ReduceArgs a{};
  int rc = fill_plan(a, plan, dtype, K, false);
  if (rc) return rc;
  GGL_REQUIRE((x || plan->E * K == 0) && (out || plan->N * K == 0), GGL_EINVAL, "x/out is NULL");
  GGL_REQUIRE((x_ld == 0 || x_ld >= K) && (out_ld == 0 || out_ld >= K), GGL_EINVAL, "row stride < K");
  a.x = x;
  a.out = out;
  a.x_ld = x_ld; a.out_ld = out_ld; a.accumulate = accumulate ? 1 : 0;
  return launch_seg<OP_SUM>(dtype, a, as_stream(stream));
}

extern "C" int ggl_segment_mean(int dtype, const void *x, const ggl_segplan_t *plan, int64_t K,
                                void *out, void *stream) {
  ReduceArgs a{};
  int rc = fill_plan(a, plan, dtype, K, false);
  if (rc) return rc;
  GGL_REQUIRE((x || plan->E * K == 0) && (out || plan->N * K == 0), GGL_EINVAL, "x/out is NULL");
  a.x = x;
  a.out = out;
  return launch_seg<OP_MEAN>(dtype, a, as_stream(stream));
}

extern "C" int ggl_segment_max(int dtype, const void *x, const ggl_segplan_t *plan, int64_t K,
                               void *out, int64_t *arg, int64_t arg_fill, void *stream) {
  ReduceArgs a{};
  int rc = fill_plan(a, plan, dtype, K, true);
  if (rc) return rc;
  GGL_REQUIRE((x || plan->E * K == 0) && ((out && arg) || plan->N * K == 0), GGL_EINVAL,
              "x/out/arg is NULL");
  // the witnesses travel in 32-bit registers (argreg_t): element positions must fit, like perm's own entries
  GGL_REQUIRE(plan->E < ((int64_t)1 << 31) && arg_fill >= 0 && arg_fill < ((int64_t)1 << 31), GGL_EINVAL,
              "segment_max: 2^31 or more elements in one plan");
  a.x = x;
  a.out = out;
  a.arg = arg;
  a.arg_fill = arg_fill;
  if (plan->E * K == 0) {
    // segment_max_cpu.cpp:28-30: an empty x returns zeros (the lowest() fill comes later), arg = fill
    if (plan->N * K > 0) {
      GGL_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)plan->N * K * dtype_size(dtype), as_stream(stream)));
      return ggl_fill_i64(arg, plan->N * K, arg_fill, stream);
    }
    return GGL_OK;
  }
  return launch_seg<OP_MAX>(dtype, a, as_stream(stream));
}

// ---- gspmm ---------------------------------------------------------------------------------------
static int spmm_common(ReduceArgs &a, const ggl_segplan_t *plan, const int32_t *col, const float *w,
                       int w_by_pos, const float *x, int64_t K, float *out, bool with_arg) {
  int rc = fill_plan(a, plan, GGL_F32, K, with_arg);
  if (rc) return rc;
  GGL_REQUIRE(col || plan->E == 0, GGL_EINVAL, "col is NULL");
  GGL_REQUIRE((x || plan->E * K == 0) && (out || plan->N * K == 0), GGL_EINVAL, "x/out is NULL");
  a.x = x;
  a.col = col;
  a.w = w;
  a.w_by_pos = w_by_pos;
  a.out = out;
  return GGL_OK;
}

extern "C" int ggl_spmm_sum(const ggl_segplan_t *plan, const int32_t *col, const float *w,
                            int w_by_pos, const float *x, int64_t K, float *out, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, K, out, false);
  if (rc) return rc;
  return launch_f32_cols<OP_SUM, MODE_SPMM>(a, as_stream(stream));
}

extern "C" int ggl_spmm_sum_ex(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                               const float *x, int64_t x_ld, int64_t K, float *out, int64_t out_ld,
                               int accumulate, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, K, out, false);
  if (rc) return rc;
  GGL_REQUIRE((x_ld == 0 || x_ld >= K) && (out_ld == 0 || out_ld >= K), GGL_EINVAL, "row stride < K");
  a.x_ld = x_ld; a.out_ld = out_ld; a.accumulate = accumulate ? 1 : 0;
  return launch_f32_cols<OP_SUM, MODE_SPMM>(a, as_stream(stream));
}

// out = dropout(relu(A x + bias)) with the epilogue applied to each finished row in registers: what
// ggl_spmm_sum followed by ggl_bias_act_fwd computes (same rounded operations, same dropout mask for the
// same rng state), minus one write and one read of [N, K].
extern "C" int ggl_spmm_sum_bias_act(const ggl_segplan_t *plan, const int32_t *col, const float *w,
                                     int w_by_pos, const float *x, int64_t K, const float *bias, int relu,
                                     float p_drop, int64_t *rng_state, float *out, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, K, out, false);
  if (rc) return rc;
  GGL_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "p_drop must be in [0, 1)");
  GGL_REQUIRE(p_drop == 0.0f || rng_state, GGL_EINVAL, "dropout needs an rng_state");
  a.epi_bias = bias;
  a.epi_rng = rng_state;
  a.epi_relu = relu ? 1 : 0;
  a.epi_thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.epi_scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  rc = launch_f32_cols<OP_SUM, MODE_SPMM_EPI>(a, as_stream(stream));
  if (rc) return rc;
  if (a.epi_thresh && plan->N > 0 && K > 0) return rng_advance(rng_state, stream);
  return GGL_OK;
}

// The general epilogue form behind ggl_spmm_sum_bias_act: sum or mean, strided x / out (a column block of a
// wider matrix), accumulate (a second edge set added onto an existing partial result: the halo-source edges of
// the multi-GPU path, whose epilogue therefore rides on the LAST block added), an extra per-row term
// `add` (SAGEConv: mean + fc_self(x_dst) + bias -> act, sage_conv.py:100-108) and the dropout word of element
// (row, epi_col0 + k) of an epi_K-wide row, so that column blocks assemble the mask of the full-width launch.
// `bias` and `add` point at the block's first column.  advance_rng != 0 steps the rng state afterwards
// (once per logical layer: set it on the last column block only).
extern "C" int ggl_spmm_epi_ex(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                               const float *x, int64_t x_ld, int64_t K, float *out, int64_t out_ld,
                               int accumulate, int mean, const float *add, int64_t add_ld, const float *bias,
                               int relu, float p_drop, int64_t *rng_state, int64_t epi_K, int64_t epi_col0,
                               int advance_rng, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, K, out, false);
  if (rc) return rc;
  GGL_REQUIRE((x_ld == 0 || x_ld >= K) && (out_ld == 0 || out_ld >= K) && (add_ld == 0 || add_ld >= K),
              GGL_EINVAL, "row stride < K");
  GGL_REQUIRE(!(mean && accumulate), GGL_EINVAL, "mean cannot accumulate onto a partial result");
  GGL_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "p_drop must be in [0, 1)");
  GGL_REQUIRE(p_drop == 0.0f || rng_state, GGL_EINVAL, "dropout needs an rng_state");
  GGL_REQUIRE(epi_col0 >= 0 && (epi_K == 0 || epi_col0 + K <= epi_K), GGL_EINVAL, "column block outside the row");
  a.x_ld = x_ld; a.out_ld = out_ld; a.accumulate = accumulate ? 1 : 0;
  a.epi_bias = bias;
  a.epi_add = add; a.add_ld = add_ld;
  a.epi_rng = rng_state;
  a.epi_relu = relu ? 1 : 0;
  a.epi_thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.epi_scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  a.epi_K = epi_K; a.epi_col0 = epi_col0;
  rc = mean ? launch_f32_cols<OP_MEAN, MODE_SPMM_EPI>(a, as_stream(stream))
            : launch_f32_cols<OP_SUM, MODE_SPMM_EPI>(a, as_stream(stream));
  if (rc) return rc;
  if (a.epi_thresh && advance_rng && plan->N > 0 && K > 0) return rng_advance(rng_state, stream);
  return GGL_OK;
}

// segment_sum / segment_mean of f32 messages x[E, K] with the same epilogue on the finished row: the
// message() + aggregate() route of a sampled SAGEConv block with "+ fc_self(x_dst) + bias -> act" fused into
// the store.  Bit-identical to ggl_segment_{sum,mean} followed by the adds in torch.
extern "C" int ggl_segment_epi(const float *x, const ggl_segplan_t *plan, int64_t K, int mean, const float *add,
                               int64_t add_ld, const float *bias, int relu, float p_drop, int64_t *rng_state,
                               float *out, void *stream) {
  ReduceArgs a{};
  int rc = fill_plan(a, plan, GGL_F32, K, false);
  if (rc) return rc;
  GGL_REQUIRE((x || plan->E * K == 0) && (out || plan->N * K == 0), GGL_EINVAL, "x/out is NULL");
  GGL_REQUIRE(add_ld == 0 || add_ld >= K, GGL_EINVAL, "row stride < K");
  GGL_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, GGL_EINVAL, "p_drop must be in [0, 1)");
  GGL_REQUIRE(p_drop == 0.0f || rng_state, GGL_EINVAL, "dropout needs an rng_state");
  a.x = x;
  a.out = out;
  a.epi_bias = bias;
  a.epi_add = add; a.add_ld = add_ld;
  a.epi_rng = rng_state;
  a.epi_relu = relu ? 1 : 0;
  a.epi_thresh = p_drop > 0.0f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.epi_scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  rc = mean ? launch_f32<OP_MEAN, MODE_SEG_EPI>(a, as_stream(stream))
            : launch_f32<OP_SUM, MODE_SEG_EPI>(a, as_stream(stream));
  if (rc) return rc;
  if (a.epi_thresh && plan->N > 0 && K > 0) return rng_advance(rng_state, stream);
  return GGL_OK;
}

extern "C" int ggl_spmm_mean(const ggl_segplan_t *plan, const int32_t *col, const float *w,
                             int w_by_pos, const float *x, int64_t K, float *out, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, K, out, false);
  if (rc) return rc;
  return launch_f32_cols<OP_MEAN, MODE_SPMM>(a, as_stream(stream));
}

extern "C" int ggl_spmm_max(const ggl_segplan_t *plan, const int32_t *col, const float *w,
                            int w_by_pos, const float *x, int64_t K, float *out, int64_t *argsrc,
                            void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, K, out, true);
  if (rc) return rc;
  GGL_REQUIRE(argsrc || plan->N * K == 0, GGL_EINVAL, "argsrc is NULL");
  a.arg = argsrc;
  a.arg_fill = 0;  // spmm_max_cpu.cpp:20: max_indices = zeros
  return launch_f32<OP_MAX, MODE_SPMM>(a, as_stream(stream));
}

extern "C" int ggl_spmm_mean_bwd(const ggl_segplan_t *planT, const int32_t *colT, const float *w,
                                 int w_by_pos, const float *g, const int64_t *fwd_rowptr, int64_t K,
                                 float *gx, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, planT, colT, w, w_by_pos, g, K, gx, false);
  if (rc) return rc;
  GGL_REQUIRE(fwd_rowptr != nullptr, GGL_EINVAL, "fwd_rowptr is NULL");
  a.aux_rowptr = fwd_rowptr;
  return launch_f32<OP_SUM, MODE_MEANBWD>(a, as_stream(stream));
}

extern "C" int ggl_spmm_max_bwd(const ggl_segplan_t *planT, const int32_t *colT, const float *w,
                                int w_by_pos, const float *g, const int64_t *argsrc, int64_t K,
                                float *gx, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, planT, colT, w, w_by_pos, g, K, gx, false);
  if (rc) return rc;
  GGL_REQUIRE(argsrc || planT->E * K == 0, GGL_EINVAL, "argsrc is NULL");
  a.aux_arg = argsrc;
  return launch_f32<OP_SUM, MODE_MAXBWD>(a, as_stream(stream));
}

extern "C" int ggl_spmm_max_bwd32(const ggl_segplan_t *planT, const int32_t *colT, const float *w,
                                  int w_by_pos, const float *g, const int32_t *argsrc32, int64_t K,
                                  float *gx, void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, planT, colT, w, w_by_pos, g, K, gx, false);
  if (rc) return rc;
  GGL_REQUIRE(argsrc32 || planT->E * K == 0, GGL_EINVAL, "argsrc32 is NULL");
  a.aux_arg = reinterpret_cast<const int64_t *>(argsrc32);
  return launch_f32<OP_SUM, MODE_MAXBWD32>(a, as_stream(stream));
}

// ---- gspmm(max) backward through a winner mask (round 5) --------------------------------------------------------------
// The backward of spmm_max_cpu.cpp:57-99 feeds w[e] * g[dst, k] to every edge e whose SOURCE is the witness stored for
// (dst, k).  Walked in source order (one output row per source, adds in ascending edge order: the reference's bits) every
// edge used to look up its destination's witness row: 8K bytes of int64 beside the 4K-byte gradient row, 3x the sum's
// traffic and 50 of the 68 ms of a K = 256 forward + backward.  The comparison only needs the witness row where it is
// WAVE-UNIFORM — in destination order: max_mask_kernel walks the forward plan, a wavefront per row (chunks of hub rows as
// in row_reduce_kernel), a lane per column holding that column's witness in a register; per edge one compare per 64
// columns IS the ballot (v_cmp writes the 64-bit lane mask), and lanes 0 .. 2 NP - 1 store the edge's K bits at its
// TRANSPOSED position (tpos: forward position -> transposed position, once per graph).  The transposed walk then streams
// K / 8 bytes per edge in its own order.  Mask: word (t * ceil(K / 32) + k / 32), bit k % 32.
template <int NP>
__global__ __launch_bounds__(kBlock) void max_mask_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const int32_t *__restrict__ tpos, const int64_t *__restrict__ argsrc,
                                                          const int32_t *__restrict__ long_rows,
                                                          const int64_t *__restrict__ chunk_ptr, uint32_t *__restrict__ mask,
                                                          int64_t N, int64_t K, int64_t KW, int64_t chunk, int64_t n_long,
                                                          int64_t n_chunks, int64_t chunk_blocks) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  int64_t row, beg, end;
  if (block_id() < chunk_blocks) {
    const int64_t cid = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(block_id() * kWavesPerBlock + wave));
    if (cid >= n_chunks) return;
    int64_t lo = 0, hi = n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= cid) lo = mid; else hi = mid - 1;
    }
    row = long_rows[lo];
    beg = rowptr[row] + (cid - chunk_ptr[lo]) * chunk;
    const int64_t rend = rowptr[row + 1];
    end = beg + chunk < rend ? beg + chunk : rend;
  } else {
    row = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((block_id() - chunk_blocks) * kWavesPerBlock + wave));
    if (row >= N) return;
    beg = rowptr[row];
    end = rowptr[row + 1];
    if (end - beg > chunk || end == beg) return;      // long rows: the chunk blocks above
  }
#ifdef GGL_EMULATE
  if (lane != 0) return;
  for (int64_t p = beg; p < end; ++p) {
    const int64_t s = col[p], t = tpos[p];
    for (int64_t wd = 0; wd < KW; ++wd) {
      uint32_t bits = 0;
      for (int b = 0; b < 32 && wd * 32 + b < K; ++b)
        if (argsrc[row * K + wd * 32 + b] == s) bits |= 1u << b;
      mask[t * KW + wd] = bits;
    }
  }
#else
  for (int64_t k0 = 0; k0 < K; k0 += (int64_t)kWave * NP) {
    int32_t a[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t k = k0 + (int64_t)kWave * i + lane;
      a[i] = k < K ? (int32_t)argsrc[row * K + k] : -1;     // (node ids are >= 0: a padding column never matches)
    }
    const int64_t w0 = k0 >> 5;
    const bool stores = lane < 2 * NP && w0 + lane < KW;
    const int half = lane & 1, plane = lane >> 1;
    auto emit = [&](int32_t s, int64_t t) {
      uint64_t b[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) b[i] = __ballot(a[i] == s);
      uint64_t mine = b[0];
#pragma unroll
      for (int i = 1; i < NP; ++i) mine = plane == i ? b[i] : mine;
      const uint32_t word = half ? (uint32_t)(mine >> 32) : (uint32_t)mine;
      if (stores) mask[t * KW + w0 + lane] = word;
    };
    int64_t p = beg;
    for (; p + 4 <= end; p += 4) {      // (unrolled by hand: the ballots are convergent operations, `#pragma unroll` declines)
      int32_t s4[4];
      int64_t t4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s4[u] = col[p + u]; t4[u] = tpos[p + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) emit(s4[u], t4[u]);
    }
    for (; p < end; ++p) emit(col[p], tpos[p]);
  }
#endif
}

// The same comparison with the mask in FORWARD position order (tpos == NULL; round 5, second form).  Scattering 126 M
// records of 8-32 bytes to transposed positions cost 6.4-7.8 ms on the products-sized graph whatever their width (every
// one a partial-line write that ends in DRAM: profiles/r5_probe_hub_and_max_backward.txt) — more than the comparison itself.
// Here the records of 32 consecutive forward positions are assembled ACROSS the wavefront's lanes — each ballot word is
// selected into the lane that owns that piece of the 32-record block — and leave as one coalesced store per 32 edges;
// the transposed walk then reads an edge's record at its forward position posT[p] (a random 8-32 byte READ beside its
// 256-1024 byte gradient row).  Record = KWp words, KWp = ggl_spmm_max_mask_words(K): ceil(K / 32) rounded up to 1, 2, 4 or
// a multiple of 8, so that a record is made of whole per-lane pieces.
// NW = words of a record this pass covers (8: two lanes x 4 words per edge; 4 / 2 / 1: one lane per edge).
#ifndef GGL_EMULATE
// lanes `lane` of r0..r3 := the wave-uniform words x0..x3 (v_writelane_b32, lane select in M0 so that the value may sit in
// any SGPR: before gfx10 the two may not be different SGPRs).  This compiler exposes no writelane builtin; the select form
// (compare on the lane id + v_cndmask per word + a v_mov per word to get the SGPR into a VGPR) costs 22 VALU instructions
// per edge at K = 256 where this costs 12.  A/B: option maxbwd_mask_wlane.
__device__ __forceinline__ void put_lane4(int lane, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t &r0, uint32_t &r1,
                                          uint32_t &r2, uint32_t &r3) {
  // M0 is a reserved register the compiler may hold a live value in (LDS-direct, movrel, sendmsg lowering) and that cannot be
  // named in a clobber list without a "may lead to undefined behaviour" diagnostic: saved and restored around the writes
  uint32_t keep;
  asm volatile("s_mov_b32 %4, m0\n\ts_mov_b32 m0, %5\n\ts_nop 1\n\tv_writelane_b32 %0, %6, m0\n\tv_writelane_b32 %1, %7, m0\n\t"
               "v_writelane_b32 %2, %8, m0\n\tv_writelane_b32 %3, %9, m0\n\ts_mov_b32 m0, %4"
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(keep)
               : "s"(lane), "s"(x0), "s"(x1), "s"(x2), "s"(x3));
}
__device__ __forceinline__ void put_lane2(int lane, uint32_t x0, uint32_t x1, uint32_t &r0, uint32_t &r1) {
  uint32_t keep;
  asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\t"
               "s_mov_b32 m0, %2"
               : "+v"(r0), "+v"(r1), "=&s"(keep)
               : "s"(lane), "s"(x0), "s"(x1));
}
#endif
template <int NP, int NW, bool WL = false>
__global__ __launch_bounds__(kBlock) void max_mask_seq_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                              const int64_t *__restrict__ argsrc,
                                                              const int32_t *__restrict__ long_rows,
                                                              const int64_t *__restrict__ chunk_ptr, uint32_t *__restrict__ mask,
                                                              int64_t N, int64_t K, int64_t KWp, int64_t chunk, int64_t n_long,
                                                              int64_t n_chunks, int64_t chunk_blocks) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  int64_t row, beg, end;
  if (block_id() < chunk_blocks) {
    const int64_t cid = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(block_id() * kWavesPerBlock + wave));
    if (cid >= n_chunks) return;
    int64_t lo = 0, hi = n_long - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (chunk_ptr[mid] <= cid) lo = mid; else hi = mid - 1;
    }
    row = long_rows[lo];
    beg = rowptr[row] + (cid - chunk_ptr[lo]) * chunk;
    const int64_t rend = rowptr[row + 1];
    end = beg + chunk < rend ? beg + chunk : rend;
  } else {
    row = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((block_id() - chunk_blocks) * kWavesPerBlock + wave));
    if (row >= N) return;
    beg = rowptr[row];
    end = rowptr[row + 1];
    if (end - beg > chunk || end == beg) return;      // long rows: the chunk blocks above
  }
#ifdef GGL_EMULATE
  if (lane != 0) return;
  for (int64_t p = beg; p < end; ++p) {
    const int64_t s = col[p];
    for (int64_t wd = 0; wd < KWp; ++wd) {
      uint32_t bits = 0;
      for (int b = 0; b < 32 && wd * 32 + b < K; ++b)
        if (argsrc[row * K + wd * 32 + b] == s) bits |= 1u << b;
      mask[p * KWp + wd] = bits;
    }
  }
#else
  for (int64_t k0 = 0; k0 < K; k0 += (int64_t)kWave * NP) {
    int32_t a[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t k = k0 + (int64_t)kWave * i + lane;
      a[i] = k < K ? (int32_t)argsrc[row * K + k] : -1;     // (node ids are >= 0: a padding column never matches)
    }
    const int64_t w0 = k0 >> 5;                              // first word of the record this pass fills
    const int e_lane = NW == 8 ? (lane >> 1) : lane;         // the edge of a 32-edge block whose piece this lane holds
    for (int64_t pb = beg & ~(int64_t)31; pb < end; pb += 32) {
      const int64_t lo = pb > beg ? pb : beg, hi = pb + 32 < end ? pb + 32 : end;
      uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      auto emit = [&](int64_t p, int32_t s) {
        const int e = (int)(p - pb);
        uint64_t b[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) b[i] = __ballot(a[i] == s);
        // the lane (NW = 8: the two lanes) that owns this edge's piece of the block takes the ballot words: a compare on the
        // lane id + one v_cndmask per word (this compiler has no v_writelane builtin; the select is two instructions more)
        if (WL && NW == 8) {
          const uint64_t bA = b[0], bB = b[NP > 1 ? 1 : 0], bC = b[NP > 2 ? 2 : 0], bD = b[NP > 3 ? 3 : 0];
          put_lane4(2 * e, (uint32_t)bA, (uint32_t)(bA >> 32), (uint32_t)bB, (uint32_t)(bB >> 32), r0, r1, r2, r3);
          put_lane4(2 * e + 1, (uint32_t)bC, (uint32_t)(bC >> 32), (uint32_t)bD, (uint32_t)(bD >> 32), r0, r1, r2, r3);
        } else if (WL && NW == 4) {
          const uint64_t bA = b[0], bB = b[NP > 1 ? 1 : 0];
          put_lane4(e, (uint32_t)bA, (uint32_t)(bA >> 32), (uint32_t)bB, (uint32_t)(bB >> 32), r0, r1, r2, r3);
        } else if (WL && NW == 2) {
          put_lane2(e, (uint32_t)b[0], (uint32_t)(b[0] >> 32), r0, r1);
        } else if (NW == 8) {
          const bool m0 = lane == 2 * e, m1 = lane == 2 * e + 1;
          const uint64_t bA = b[0], bB = b[NP > 1 ? 1 : 0], bC = b[NP > 2 ? 2 : 0], bD = b[NP > 3 ? 3 : 0];
          r0 = m0 ? (uint32_t)bA : m1 ? (uint32_t)bC : r0;
          r1 = m0 ? (uint32_t)(bA >> 32) : m1 ? (uint32_t)(bC >> 32) : r1;
          r2 = m0 ? (uint32_t)bB : m1 ? (uint32_t)bD : r2;
          r3 = m0 ? (uint32_t)(bB >> 32) : m1 ? (uint32_t)(bD >> 32) : r3;
        } else {
          const bool m0 = lane == e;
          r0 = m0 ? (uint32_t)b[0] : r0;
          if (NW >= 2) r1 = m0 ? (uint32_t)(b[0] >> 32) : r1;
          if (NW >= 4) {
            r2 = m0 ? (uint32_t)b[NP > 1 ? 1 : 0] : r2;
            r3 = m0 ? (uint32_t)(b[NP > 1 ? 1 : 0] >> 32) : r3;
          }
        }
      };
      int64_t p = lo;
      for (; p + 4 <= hi; p += 4) {      // (unrolled by hand: ballots are convergent operations, `#pragma unroll` declines)
        int32_t s4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = col[p + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) emit(p + u, s4[u]);
      }
      for (; p < hi; ++p) emit(p, col[p]);
      // one store per lane and 32-edge block: the lanes whose edge lies in [lo, hi) — a block shared with the neighbouring
      // row (or chunk) is completed by that row's wavefront, each lane's piece belongs to exactly one edge
      const int64_t pe = pb + e_lane;
      if (pe >= lo && pe < hi && (NW == 8 || lane < 32)) {
        uint32_t *dst = mask + pe * KWp + w0 + (NW == 8 ? 4 * (lane & 1) : 0);
        if (NW >= 4) *reinterpret_cast<uint4 *>(dst) = uint4{r0, r1, r2, r3};
        else if (NW == 2) *reinterpret_cast<uint2 *>(dst) = uint2{r0, r1};
        else *dst = r0;
      }
    }
  }
#endif
}

// words per edge record: transposed-order mask (scatter form) ceil(K / 32); forward-order mask ggl_spmm_max_mask_words(K)
static int64_t mask_words_seq(int64_t K) {
  const int64_t kw = (K + 31) / 32;
  return kw <= 1 ? 1 : kw <= 2 ? 2 : kw <= 4 ? 4 : ((kw + 7) / 8) * 8;
}
extern "C" int64_t ggl_spmm_max_mask_words(int64_t K, int forward_order) {
  return forward_order ? mask_words_seq(K) : (K + 31) / 32;
}
extern "C" size_t ggl_spmm_max_mask_bytes(int64_t E, int64_t K) {     // (room for either form)
  return (size_t)(E > 0 ? E : 0) * (size_t)mask_words_seq(K) * sizeof(uint32_t);
}

// Which form the gspmm(max) backward takes (include/ggl_mpops.h): 2 = 1-bit winner mask, 1 = int32 witness copy, 0 = the
// int64 witnesses as they are.  The mask is a TRANSIENT of E x mask_words_seq(K) x 4 bytes: taken for maxbwd_mask <= K <=
// maxbwd_mask_kmax (128 .. 256: at most 32 bytes per edge, twice what the plan itself holds per edge).  Measured on both
// benchmark graphs (profiles/r6_maxbwd_forms.txt, fwd + bwd, int32 witnesses vs mask): products-sized K = 128 / 256 / 602:
// 24.8 / 52.9 / 193.5 vs 21.8 / 41.4 / 213.6 ms; Reddit-sized (dense: its 0.48 GB witness matrix is cache-resident) 16.5 /
// 40.8 / 136.9 vs 15.3 / 30.5 / 159.9 ms — the mask wins up to K = 256 on both and loses at K = 602 on both (96 B per edge:
// +11.9 GiB on the Reddit-sized graph).  The hosts fall back to form 1 when the transient cannot be allocated.
extern "C" int ggl_policy_maxbwd_form(int64_t E, int64_t N_dst, int64_t K) {
  const auto &o = options();
  (void)N_dst;
  if (o.maxbwd_mask > 0 && K >= o.maxbwd_mask && (o.maxbwd_mask_kmax <= 0 || K <= o.maxbwd_mask_kmax) && E > 0) return 2;
  return o.maxbwd_arg32 != 0 ? 1 : 0;
}

extern "C" int ggl_spmm_max_mask(const ggl_segplan_t *planF, const int32_t *colF, const int32_t *tpos,
                                 const int64_t *argsrc, int64_t K, uint32_t *mask, void *stream) {
  GGL_REQUIRE(planF != nullptr && planF->rowptr != nullptr, GGL_EINVAL, "plan is NULL");
  const int64_t E = planF->E, N = planF->N;
  if (E <= 0 || K <= 0 || N <= 0) return GGL_OK;
  GGL_REQUIRE(colF && argsrc && mask, GGL_EINVAL, "ggl_spmm_max_mask: NULL argument");
  GGL_REQUIRE(N < ((int64_t)1 << 31), GGL_EINVAL, "too many rows");
  GGL_REQUIRE(planF->n_long == 0 || (planF->long_rows && planF->chunk_ptr), GGL_EINVAL, "plan has long rows but no chunk list");
  const int64_t chunk_blocks = planF->n_long > 0 ? ceil_div(planF->n_chunks, (int64_t)kWavesPerBlock) : 0;
  const int64_t grid = chunk_blocks + ceil_div(N, (int64_t)kWavesPerBlock);
  hipStream_t st = as_stream(stream);
  if (tpos == nullptr) {      // forward-order records, assembled across lanes, one coalesced store per 32 edges
    const int64_t KWp = mask_words_seq(K);
    GGL_REQUIRE((reinterpret_cast<uintptr_t>(mask) & 15u) == 0, GGL_EINVAL, "mask must be 16-byte aligned");
#ifdef GGL_EMULATE
    constexpr bool kCanWl = false;
#else
    constexpr bool kCanWl = true;
#endif
    const bool wl = kCanWl && options().maxbwd_mask_wlane != 0;
#define GGL_MS(NP, NW)                                                                                              \
  do {                                                                                                              \
    if (wl) GGL_LAUNCH((max_mask_seq_kernel<NP, NW, kCanWl>), grid, kBlock, st, planF->rowptr, colF, argsrc, planF->long_rows, \
                       planF->chunk_ptr, mask, N, K, KWp, planF->chunk, planF->n_long, planF->n_chunks, chunk_blocks);       \
    else GGL_LAUNCH((max_mask_seq_kernel<NP, NW, false>), grid, kBlock, st, planF->rowptr, colF, argsrc, planF->long_rows,  \
                    planF->chunk_ptr, mask, N, K, KWp, planF->chunk, planF->n_long, planF->n_chunks, chunk_blocks);          \
  } while (0)
    if (K <= 32) GGL_MS(1, 1);
    else if (K <= 64) GGL_MS(1, 2);
    else if (K <= 128) GGL_MS(2, 4);
    else GGL_MS(4, 8);
#undef GGL_MS
    GGL_LAUNCH_CHECK();
    return GGL_OK;
  }
  const int64_t KW = (K + 31) / 32;
#define GGL_MM(NP)                                                                                                  \
  GGL_LAUNCH((max_mask_kernel<NP>), grid, kBlock, st, planF->rowptr, colF, tpos, argsrc, planF->long_rows,         \
             planF->chunk_ptr, mask, N, K, KW, planF->chunk, planF->n_long, planF->n_chunks, chunk_blocks)
  if (K <= 64) GGL_MM(1);
  else if (K <= 128) GGL_MM(2);
  else GGL_MM(4);
#undef GGL_MM
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}

extern "C" int ggl_spmm_max_bwd_mask(const ggl_segplan_t *planT, const int32_t *colT, const float *w, int w_by_pos,
                                     const float *g, const uint32_t *mask, const int32_t *mask_pos, int64_t K, float *gx,
                                     void *stream) {
  ReduceArgs a{};
  int rc = spmm_common(a, planT, colT, w, w_by_pos, g, K, gx, false);
  if (rc) return rc;
  GGL_REQUIRE(mask || planT->E * K == 0, GGL_EINVAL, "mask is NULL");
  a.aux_arg = reinterpret_cast<const int64_t *>(mask);
  // mask_pos = posT (the forward position of every transposed position): records in forward order; NULL: in transposed order
  a.aux_rowptr = reinterpret_cast<const int64_t *>(mask_pos);
  a.mask_words = mask_pos ? mask_words_seq(K) : (K + 31) / 32;
  // ONE launch over the full width: the 64-column blocks that pay for the plain sum re-read every edge's mask record (and
  // posT entry) once per block — measured on the products-sized graph at K = 256: the scattered-record walk 16.5 ms in one
  // launch, 21.5 in four; with forward-order records (a random 32-byte read per edge and block) 32 ms
  // (profiles/r5_max_backward.txt).  GGL_MAXBWD_MASK_COLS=1 takes the blocks (A/B).
  if (options().maxbwd_mask_cols != 0) return launch_f32_cols<OP_SUM, MODE_MAXBWDM>(a, as_stream(stream));
  return launch_f32<OP_SUM, MODE_MAXBWDM>(a, as_stream(stream));
}

// ---- bspmm ---------------------------------------------------------------------------------------
extern "C" int ggl_bspmm_sum(const ggl_segplan_t *plan, const int32_t *col, const float *w,
                             int w_by_pos, const float *x, int64_t H, int64_t C, float *out,
                             void *stream) {
  ReduceArgs a{};
  GGL_REQUIRE(H > 0 && C > 0, GGL_EINVAL, "H and C must be positive");
  int rc = spmm_common(a, plan, col, w, w_by_pos, x, H * C, out, false);
  if (rc) return rc;
  a.H = H;
  a.C = C;
  return launch_bspmm_cols(a, as_stream(stream));
}

// ---- timing aid for bench.py's roofline leg ------------------------------------------------------
extern "C" int ggl_time_spmm_sum(const ggl_segplan_t *plan, const int32_t *col, const float *w,
                                 int w_by_pos, const float *x, int64_t K, float *out, void *stream,
                                 int reps, float *ms_host) {
#ifdef GGL_EMULATE
  (void)plan; (void)col; (void)w; (void)w_by_pos; (void)x; (void)K; (void)out; (void)stream; (void)reps;
  *ms_host = 0.0f;
  return GGL_OK;
#else
  GGL_REQUIRE(reps > 0 && ms_host, GGL_EINVAL, "reps must be > 0");
  hipStream_t s = as_stream(stream);
  hipEvent_t e0, e1;
  GGL_HIP_CHECK(hipEventCreate(&e0));
  GGL_HIP_CHECK(hipEventCreate(&e1));
  int rc = ggl_spmm_sum(plan, col, w, w_by_pos, x, K, out, stream);  // warm
  if (rc) return rc;
  GGL_HIP_CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) {
    rc = ggl_spmm_sum(plan, col, w, w_by_pos, x, K, out, stream);
    if (rc) return rc;
  }
  GGL_HIP_CHECK(hipEventRecord(e1, s));
  GGL_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  GGL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_host = ms / (float)reps;
  GGL_HIP_CHECK(hipEventDestroy(e0));
  GGL_HIP_CHECK(hipEventDestroy(e1));
  return GGL_OK;
#endif
}
