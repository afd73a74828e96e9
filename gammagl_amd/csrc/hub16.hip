// gammagl_amd/csrc/hub16.hip — hub rows of the 16-bit float segment sums (GPU build only: LDS + barriers).
//
// f16 / bf16 sums accumulate in the storage type (segment_sum_cpu.cpp:47-56: the running sum is rounded to 16 bits
// after every add, so a sum of ones sticks at 2048): the result depends on the serial order far beyond rounding and a
// hub row cannot be cut into independently reduced chunks.  The row kernel of reduce.hip therefore walks such a row
// with ONE lane group, four dependent gathers at a time — 0.2-0.3 us per element, 12-32 ms for a 109 110-element hub
// where the f32 op (chunked) takes 0.15-0.7 ms.  What is serial is only the ADD chain, not the loads: here a
// workgroup owns (hub row, 64-column slab); four producer wavefronts gather the next 256 elements of the slab
// through `perm` into one half of a double-buffered LDS tile (one element per thread, up to eight independent
// 16-byte loads in flight each, the index fetched a stage earlier) while the consumer wavefront — one column per lane — folds the other half into its running sums in the
// reference's order: load, add in f32, round to the storage type (3 VALU instructions per element).  Bit-identical to
// the serial walk; the slabs of a row and the hub rows run in parallel.
#include "common.hpp"

namespace ggl {

// the stage barrier: only the LDS tile is shared, so the fences name the LDS address space alone — __syncthreads() is a
// full workgroup-scope release, which on gfx9 waits for EVERY outstanding memory operation of the wave (vmcnt(0)): the
// producers' index prefetch of the next stage drained at every stage (round 4, see hubf32.hip)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int kHubStage = 256;              // elements per LDS half = producer threads (one element each)
constexpr int kHubCols = 64;                // columns per slab = consumer lanes
constexpr int kHubBlock = kWave + kHubStage;  // wavefront 0 consumes, wavefronts 1-4 produce

// The consumer's running sum: ONE dependent chain per column, so its length per element is the kernel's floor.
// TT<T>::add spells "add in f32, round to the storage type, widen again" with software rounding (6-7 dependent
// VALU operations for bf16).  Same values from shorter chains:
//   f16 : v_add_f16 on the storage values.  The sum of two f16 numbers rounded once to f16 equals the f32 sum rounded
//         to f16 — double rounding is innocuous when the wider format has at least 2p + 2 significand bits
//         (p = 11: 24 >= 24) — so this is c10::Half's "float add, then round" bit for bit (1 instruction per element);
//   bf16: f32 add, v_cvt_pk_bf16_f32 (round to nearest even in hardware, gfx950), shift back: 3 instructions.  A NaN stays
//         a NaN along the chain whatever its payload; the final store canonicalises it like c10::BFloat16 (0x7FC0).
template <typename T> struct HubAcc;
template <> struct HubAcc<f16_t> {
  _Float16 a = (_Float16)0.0f;
  __device__ __forceinline__ void add(uint16_t xb) {
    _Float16 x;
    __builtin_memcpy(&x, &xb, 2);
    a = a + x;
  }
  __device__ __forceinline__ float value() const { return (float)a; }
};
template <> struct HubAcc<bf16_t> {
  float a = 0.0f;
  __device__ __forceinline__ void add(uint16_t xb) {
    const __bf16 r = (__bf16)__fadd_rn(a, bf16_to_f32(xb));
    uint16_t bits;
    __builtin_memcpy(&bits, &r, 2);
    a = bf16_to_f32(bits);
  }
  __device__ __forceinline__ float value() const { return a; }   // (TT<bf16_t>::store maps a NaN to 0x7FC0)
};

// One element's slab (ncol <= 64 columns starting at column c0 of its row) from memory into its LDS row: P whole
// 8-element pieces as 16-byte loads — all in flight together, P a compile-time constant (a run-time bound on the piece
// loop sent the staging array to scratch memory: 144 bytes per thread in round 2's kernel) — plus, for rows that are not
// made of aligned 16-byte pieces (V16 = false: K = 47 -> 94-byte rows, or an unaligned base), the ncol % 8 elements left
// over as the tail of the 16 bytes that END at the slab's end.  That path used to issue one 2-byte load per element
// (47 per row: the K = 47 hub rows of the products-sized graph took 16 ms while the other 2.4 M rows took 8).
template <bool V16, int P>
__device__ __forceinline__ void hub_fill(const uint16_t *__restrict__ g, uint16_t *__restrict__ dst, int ncol, int64_t c0) {
  uint4 v[P > 0 ? P : 1];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    if (V16) {
      v[q] = reinterpret_cast<const uint4 *>(g)[q];
    } else {
      const H8U t = *reinterpret_cast<const H8U *>(g + q * 8);     // 2-byte aligned 16-byte load
      __builtin_memcpy(&v[q], &t, 16);
    }
  }
  uint64_t lo = 0, hi = 0;
  const int rem = V16 ? 0 : (ncol & 7);
  const bool shifted = rem > 0 && (c0 + ncol >= 8);               // (always, unless the whole row is shorter than 8)
  if (shifted) {
    const H8U t = *reinterpret_cast<const H8U *>(g + ncol - 8);
    uint4 tw;
    __builtin_memcpy(&tw, &t, 16);
    lo = (uint64_t)tw.x | ((uint64_t)tw.y << 32);
    hi = (uint64_t)tw.z | ((uint64_t)tw.w << 32);
  }
#pragma unroll
  for (int q = 0; q < P; ++q) *reinterpret_cast<uint4 *>(dst + q * 8) = v[q];
  if (rem > 0) {
#pragma unroll
    for (int t = 0; t < 7; ++t)
      if (t < rem) {
        const int j = 8 - rem + t;                                 // element j of the 8 loaded (shifts, no register indexing)
        dst[P * 8 + t] = shifted ? (uint16_t)((j < 4 ? lo : hi) >> (16 * (j & 3))) : g[P * 8 + t];
      }
  }
}

// V16: K % 8 == 0 and 16-byte aligned rows — the slab of an element moves as 16-byte pieces; otherwise element by element
template <typename T, bool V16>
__global__ __launch_bounds__(kHubBlock) void hub_rows16_kernel(const uint16_t *__restrict__ x,
                                                               const int32_t *__restrict__ perm,
                                                               const int64_t *__restrict__ rowptr,
                                                               const int32_t *__restrict__ long_rows,
                                                               const int32_t *__restrict__ long_order, int64_t n_long,
                                                               int64_t K, int64_t slabs, int mean,
                                                               uint16_t *__restrict__ out) {
  __shared__ uint16_t buf[2][kHubStage][kHubCols];   // 2 x 32 KiB
  const int64_t jb = block_id() / slabs, slab = block_id() - jb * slabs;
  if (jb >= n_long) return;
  const int64_t row = long_rows[long_order ? (int64_t)long_order[jb] : jb];   // longest row first (ggl_segplan.long_order)
  const int64_t beg = rowptr[row], end = rowptr[row + 1], len = end - beg;
  const int64_t c0 = slab * kHubCols;
  const int ncol = (int)((K - c0) < kHubCols ? (K - c0) : kHubCols);   // (V16: a multiple of 8)
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t nst = (len + kHubStage - 1) / kHubStage;
  const int parts = ncol >> 3;                                          // 16-byte pieces per element of this slab
  if (tid >= kWave) {
    // ---- producers: thread e owns element e of every stage; its source row index is fetched one stage ahead of
    // the row itself, so a stage costs one gather latency, not two
    const int e = tid - kWave;
    auto src_of = [&](int64_t st) -> int64_t {
      const int64_t p = beg + st * kHubStage + e;
      return p < end ? (perm ? (int64_t)perm[p] : p) : -1;
    };
    auto fill = [&](int64_t src, int b) {
      if (src < 0) return;
      const uint16_t *g = x + src * K + c0;
      uint16_t *dst = &buf[b][e][0];
      switch (parts) {   // block-uniform: the piece count is a compile-time constant inside each case (see hub_fill)
        case 8: hub_fill<V16, 8>(g, dst, ncol, c0); break;
        case 7: hub_fill<V16, 7>(g, dst, ncol, c0); break;
        case 6: hub_fill<V16, 6>(g, dst, ncol, c0); break;
        case 5: hub_fill<V16, 5>(g, dst, ncol, c0); break;
        case 4: hub_fill<V16, 4>(g, dst, ncol, c0); break;
        case 3: hub_fill<V16, 3>(g, dst, ncol, c0); break;
        case 2: hub_fill<V16, 2>(g, dst, ncol, c0); break;
        case 1: hub_fill<V16, 1>(g, dst, ncol, c0); break;
        default: hub_fill<V16, 0>(g, dst, ncol, c0); break;
      }
    };
    int64_t src = src_of(0);
    int64_t nxt = nst > 1 ? src_of(1) : -1;
    fill(src, 0);
    lds_barrier();
    for (int64_t st = 0; st < nst; ++st) {
      src = nxt;
      nxt = st + 2 < nst ? src_of(st + 2) : -1;
      if (st + 1 < nst) fill(src, (int)((st + 1) & 1));
      lds_barrier();
    }
    return;
  }
  // ---- consumer: one column per lane, the elements of a stage in order; the LDS reads of 8 elements are issued
  // together, the adds stay the reference's serial chain
  HubAcc<T> run;
  lds_barrier();
  for (int64_t st = 0; st < nst; ++st) {
    const int b = (int)(st & 1);
    if (lane < ncol) {
      const int cnt = (int)((len - st * kHubStage) < kHubStage ? (len - st * kHubStage) : kHubStage);
      int e = 0;
      for (; e + 8 <= cnt; e += 8) {
        uint16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = buf[b][e + q][lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) run.add(v[q]);
      }
      for (; e < cnt; ++e) run.add(buf[b][e][lane]);
    }
    lds_barrier();
  }
  typename TT<T>::A acc = run.value();
  if (lane < ncol) {
    if (mean) {  // segment_mean_cpu.cpp:67-76: the count lives in the storage type; divide only where count > 1
      const typename TT<T>::A c = TT<T>::count(len);
      if (TT<T>::gt1(c)) acc = TT<T>::div(acc, c);
    }
    out[row * K + c0 + lane] = TT<T>::store(acc);
  }
}

}  // namespace ggl

using namespace ggl;

extern "C" int ggl_segment_hub16_supported(int dtype, int64_t K, const void *x, const void *out) {
  return ((dtype == GGL_F16 || dtype == GGL_BF16) && K > 0 && (reinterpret_cast<uintptr_t>(x) & 1u) == 0 &&
          (reinterpret_cast<uintptr_t>(out) & 1u) == 0) ? 1 : 0;
}

// sum (mean = 0) / mean (mean = 1) of the plan's LONG rows only (plan->long_rows, n_long), each in the reference's
// serial order.  Pair it with ggl_segment_{sum,mean} on the same plan with the long-row table withheld (n_long = 0,
// chunk kept): that launch skips the rows longer than chunk, this one fills them in.
extern "C" int ggl_segment_hub16(int dtype, int mean, const void *x, const ggl_segplan_t *plan, int64_t K, void *out,
                                 void *stream) {
  GGL_REQUIRE(plan && plan->rowptr, GGL_EINVAL, "plan is NULL");
  GGL_REQUIRE(ggl_segment_hub16_supported(dtype, K, x, out), GGL_EINVAL,
              "ggl_segment_hub16: f16 / bf16 rows");
  if (plan->n_long <= 0) return GGL_OK;
  GGL_REQUIRE(plan->long_rows && x && out, GGL_EINVAL, "NULL pointer");
  const int64_t slabs = ceil_div(K, (int64_t)kHubCols);
  const int64_t grid = plan->n_long * slabs;
  hipStream_t s = as_stream(stream);
  const bool v16 = K % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
#define GGL_HUB16(T, V)                                                                                          \
  GGL_LAUNCH((hub_rows16_kernel<T, V>), grid, kHubBlock, s, static_cast<const uint16_t *>(x), plan->perm, plan->rowptr, \
             plan->long_rows, plan->long_order, plan->n_long, K, slabs, mean ? 1 : 0, static_cast<uint16_t *>(out))
  if (dtype == GGL_F16) {
    if (v16) GGL_HUB16(f16_t, true); else GGL_HUB16(f16_t, false);
  } else {
    if (v16) GGL_HUB16(bf16_t, true); else GGL_HUB16(bf16_t, false);
  }
#undef GGL_HUB16
  GGL_LAUNCH_CHECK();
  return GGL_OK;
}
