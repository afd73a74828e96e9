// gammagl_amd/csrc/hubf32.hip — hub rows of the f32 (and f64 segment) sums IN THE REFERENCE'S SERIAL ORDER (GPU build only: LDS + barriers).
//
// The reference adds a row's elements one after the other (spmm_sum_cpu.cpp:29-39: for e in edge order:
// out[dst] += w[e] * x[src]; segment_sum_cpu.cpp:47-56 likewise), every add rounded to f32.  The row kernel of
// reduce.hip reproduces that chain add for add on rows it walks in one piece; rows LONGER than the plan's chunk used to
// be cut into chunks reduced by independent wavefronts and combined in chunk order — the same adds in another
// association, i.e. a result within rounding of the reference's (measured against the reference's own c_spmm_sum at the
// products size: 1752 of 2 449 029 rows differ, by up to 1.4e-5 of the row's magnitude) but not its bits.
//
// What is serial in such a row is only the ADD chain, not the gathers that feed it.  Here a workgroup owns
// (hub row, 64-column slab): eight producer wavefronts gather the row's elements — four per load instruction, 16 lanes x
// 16 bytes each, four stages (up to 512 elements) in flight per workgroup — multiply them by their edge weight (one rounded
// multiply, as the row kernel does) and park them in a double-buffered LDS tile; ONE consumer wavefront, a column per
// lane, folds the tile into its running sum in element order: ds_read_b32 + v_add_f32, eight reads issued ahead of the
// eight dependent adds.  Its output is one partial row per hub row, which long_final_kernel (reduce.hip) turns into the
// output row exactly like a chunk partial (mean, bias / ReLU / dropout epilogue, accumulate) — so every mode that sums
// goes through here unchanged.  The launch runs on a side stream BESIDE the launch over the other rows (disjoint
// outputs), forked and joined with events inside the library (legal under hipGraph capture).
//
// Cost model: the chain is >= len x ~8 cycles (a 150 000-element hub: 0.5 ms); a workgroup sustains ~384 elements per
// gather latency.  The bytes are the ones the chunked walk moved (each element's slab once).
#include "common.hpp"

#include <mutex>

namespace ggl {

constexpr int kHfProd = 8;                           // producer wavefronts
constexpr int kHfDepth = 4;                          // stages of gathers in flight per producer (and of ids ahead of them)
constexpr int kHfBlock = kWave * (1 + kHfProd);      // wavefront 0 consumes
// PER = load instructions per producer lane and stage (4 elements each): a stage is 32 PER elements.
//   PER = 4: 128-element stages, 512 elements in flight, 144 VGPRs + 64 KiB of LDS -> one workgroup per CU: the
//            configuration for LONG hub rows (products-sized graph, chunk 4096: 14.81 ms per K = 256 aggregate against
//            14.78 chunked; PER = 2 there: 15.55);
//   PER = 2:  64-element stages, 78 VGPRs + 32 KiB -> two workgroups per CU and half the padding of a short row: the
//            configuration where "long" starts at 257 elements (arxiv-sized graph: 0.347 ms against 0.386 with PER = 4;
//            chunked 0.290) — profiles/r4_hub_exact_timing.txt.
constexpr int64_t kHfHeavyFrom = 4096;               // average long-row length from which PER = 4 is launched
// LOG_LPE = log2 of the lanes that move one element's slab: 4 -> 16 lanes x 16 bytes = 64-column slabs, four elements per
// load instruction; 2 -> 4 lanes = 16-column slabs, SIXTEEN elements per load instruction — rows of K <= 16 columns
// (class scores, attention logits, degree-like sums), where a 64-column slab left 14 of every 16 producer lanes idle:
// gspmm K = 7 on the products-sized graph 1.98 ms chunked -> 3.01 ms with 64-column slabs -> see profiles/r4_hub_exact_timing.txt.

// The stage barrier.  __syncthreads() is a workgroup-scope release + acquire fence around s_barrier, and on gfx9 the
// release waits for EVERY outstanding memory operation of the wave (s_waitcnt vmcnt(0)) — the producers' gathers of the
// next stages included: with it each stage cost one full memory round trip however deep the pipeline was (2.15 us per
// 128-element stage measured, profiles/r4_hub_exact_timeline.txt).  Only the LDS tile is shared here, so the fences
// name the LDS address space alone: the barrier waits for the wave's LDS traffic (lgkmcnt) and nothing else.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One pipeline slot of a producer lane: the ids / weights of a stage (requested kHfDepth iterations before its rows are)
// and the rows themselves (requested kHfDepth iterations before they are parked).  The memory counter (vmcnt) retires
// loads IN ORDER, so a wait for any load waits for every older one: ids and rows of the same slot are therefore issued
// back to back, `rows(s), ids(s + Depth)`, and by the time either is needed everything issued after them — three more
// stages of rows and ids — may still be in flight.  For the backend to COUNT that (s_waitcnt vmcnt(n), n > 0) the loop
// body must be straight-line code: the index mode is a template parameter and no load is conditional — positions past
// the row's end and lanes past the slab's last column load a clamped (valid) address and are simply not used.
enum HubMode {
  HUB_SEG = 0,        // rows x[p]                      (ids arrived sorted)
  HUB_SEG_PERM = 1,   // rows x[perm[p]]
  HUB_SPMM = 2,       // rows x[col[p]], no weights
  HUB_SPMM_W = 3,     // ... * w[p]                     (weights in sorted order: every call but the first of a weight vector)
  HUB_SPMM_WP = 4,    // ... * w[perm[p]]               (first sight: a dependent load, drains the pipeline — once)
  HUB_BSPMM_W = 5,    // ... * w[p, head(column)]
  HUB_BSPMM_WP = 6    // ... * w[perm[p], head(column)]
};
constexpr bool hub_seg(int m) { return m == HUB_SEG || m == HUB_SEG_PERM; }
constexpr bool hub_has_w(int m) { return m >= HUB_SPMM_W; }
constexpr bool hub_w_perm(int m) { return m == HUB_SPMM_WP || m == HUB_BSPMM_WP; }
constexpr bool hub_heads(int m) { return m == HUB_BSPMM_W || m == HUB_BSPMM_WP; }

template <int PER> struct HfRows { float4 r[PER]; };                 // the gathered quads of one stage
template <int PER, int NW> struct HfW { float w[PER][NW]; };          // their weights (NW = 4: one per column — heads that change inside a quad)
template <int PER> struct HfIds { int32_t row[PER], wi[PER]; };       // source rows (+ weight positions when they go through perm) of a stage

// VEC4: K % 4 == 0, 16-byte aligned base and row stride — a lane moves its four columns as one 16-byte load.
// WPC (multi-head weights only): the head changes inside a quad (C % 4 != 0) — a weight per column.
// F64 (segment sums of doubles: the two segment modes, no weights): the producers move the row as 4-byte words — a row of
// K doubles is a row of 2 K words, `a.K` / `a.x_ld` / `a.x` arrive in words — and only the consumer knows better: a lane
// owns a DOUBLE (two adjacent words of the tile) and adds with __dadd_rn.  Same pipeline, same order of adds.
template <int MODE, bool VEC4, bool WPC, int PER, int LOG_LPE, bool F64 = false, bool PIPE = true>
__global__ __launch_bounds__(kHfBlock) void hub_rows_f32_kernel(const HubF32Args a) {
  static_assert(!F64 || hub_seg(MODE), "doubles: segment sums only (the SpMMs are f32 in the reference)");
  constexpr int kLpe = 1 << LOG_LPE, kEpl = kWave / kLpe;          // lanes per element, elements per load instruction
  constexpr int kHfCols = kLpe * 4;                                 // columns per slab
  constexpr int kHfPer = PER, kHfStage = kHfProd * PER * kEpl;      // elements per LDS half
  __shared__ __attribute__((aligned(16))) float buf[2][kHfStage][kHfCols];   // 2 x 32 KiB (PER = 4, 64 columns) ... 2 x 16 KiB
  constexpr int NW = WPC ? 4 : 1;
  const int64_t slabs = (a.K + kHfCols - 1) / kHfCols;
  const int64_t jb = block_id() / slabs, slab = block_id() - jb * slabs;
  if (jb >= a.n_long) return;
  // the grid is walked longest row first (ggl_segplan.long_order): the longest add chain bounds the launch, so it starts
  // with the first workgroups; its partial row still sits at its position j in long_rows
  const int64_t j = a.long_order ? (int64_t)a.long_order[jb] : jb;
  const int64_t row = a.long_rows[j];
  const int64_t beg = a.rowptr[row], end = a.rowptr[row + 1], len = end - beg;   // (a long row: len > 0)
  const int64_t c0 = slab * kHfCols;
  const int ncol = (int)((a.K - c0) < kHfCols ? (a.K - c0) : kHfCols);
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t nst = (len + kHfStage - 1) / kHfStage;
  const int64_t nstp = (nst + 3) & ~(int64_t)3;      // stages incl. the padding of the last group of four (see below)
  if (tid >= kWave) {
    // ---- producers ------------------------------------------------------------------------------------------------
    const int pw = (tid >> 6) - 1, eg = lane >> LOG_LPE, piece = lane & (kLpe - 1);
    const bool live = piece * 4 < ncol;
    const int cl = live ? piece * 4 : 0;                        // first of this lane's four columns inside the slab
    const int e0 = pw * (kHfPer * kEpl) + eg;                   // this lane's elements of a stage: e0 + kEpl i
    const float *xs = a.x + c0 + cl;
    int64_t hd[NW];                                             // head of each weight this lane applies
#pragma unroll
    for (int c = 0; c < NW; ++c) {
      const int64_t k = c0 + cl + c;
      hd[c] = hub_heads(MODE) ? (k < a.K ? k : a.K - 1) / a.C : 0;
    }
    // register rings of Depth slots each (slot = stage % Depth), nothing is ever copied between slots:
    //   step s:  park stage s (R, W);  request rows + weights of stage s + Depth into the slots just freed, addressed by
    //            the ids requested at step s - Depth;  request the ids of stage s + 2 Depth into the id slot just used.
    HfRows<PER> R[kHfDepth];
    HfW<PER, NW> W[kHfDepth];
    HfIds<PER> I[kHfDepth];
    auto load_idx = [&](HfIds<PER> &t, int64_t st) {                  // ids of stage st (positions clamped into the row)
#pragma unroll
      for (int i = 0; i < kHfPer; ++i) {
        int64_t p = beg + st * kHfStage + e0 + kEpl * i;
        p = p < end ? p : end - 1;
        if (MODE == HUB_SEG) t.row[i] = (int32_t)p;
        else if (MODE == HUB_SEG_PERM) t.row[i] = a.perm[p];
        else t.row[i] = a.col[p];
        if (hub_has_w(MODE)) t.wi[i] = hub_w_perm(MODE) ? a.perm[p] : (int32_t)p;
      }
    };
    auto issue = [&](HfRows<PER> &d, HfW<PER, NW> &wv, const HfIds<PER> &t) {   // rows + weights of the stage whose ids sit in t
#pragma unroll
      for (int i = 0; i < kHfPer; ++i) {
        const float *g = xs + (int64_t)t.row[i] * a.x_ld;
        if (VEC4) {
          d.r[i] = *reinterpret_cast<const float4 *>(g);
        } else {                                                  // (columns past the slab's end: the last valid one again)
          const int last = ncol - 1 - cl;
          d.r[i].x = g[0];
          d.r[i].y = g[last < 1 ? last : 1];
          d.r[i].z = g[last < 2 ? last : 2];
          d.r[i].w = g[last < 3 ? last : 3];
        }
        if (hub_has_w(MODE)) {
#pragma unroll
          for (int c = 0; c < NW; ++c)
            wv.w[i][c] = hub_heads(MODE) ? a.w[(int64_t)t.wi[i] * a.H + hd[c]] : a.w[t.wi[i]];
        }
      }
    };
    auto park = [&](const HfRows<PER> &d, const HfW<PER, NW> &wv, int b) {
#pragma unroll
      for (int i = 0; i < kHfPer; ++i) {
        float4 v = d.r[i];
        if (hub_has_w(MODE)) {
          v.x = __fmul_rn(wv.w[i][0], v.x);
          v.y = __fmul_rn(wv.w[i][WPC ? 1 : 0], v.y);
          v.z = __fmul_rn(wv.w[i][WPC ? 2 : 0], v.z);
          v.w = __fmul_rn(wv.w[i][WPC ? 3 : 0], v.w);
        }
        // every lane parks at its own columns: elements past the row's end and columns past the slab's end are never read
        *reinterpret_cast<float4 *>(&buf[b][e0 + kEpl * i][piece * 4]) = v;
      }
    };
    static_assert(kHfDepth == 4, "the stage loop below is unrolled for four slots");
    // (sched_barrier: the backend's scheduler must not reorder the gathers across slots — it hoisted slot 0's rows to
    //  the END of the prologue, after which every iteration waited for all outstanding loads: vmcnt(0) in the ISA)
    // prologue: the ids of stages 0 .. Depth-1, then per slot `rows + weights (k), ids (k + Depth)`
#pragma unroll
    for (int k = 0; k < kHfDepth; ++k) load_idx(I[k], k);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kHfDepth; ++k) {
      issue(R[k], W[k], I[k]);
      __builtin_amdgcn_sched_barrier(0);
      load_idx(I[k], k + kHfDepth);
      __builtin_amdgcn_sched_barrier(0);
    }
#define GGL_HF_STEP(SL, ST)                                   \
    park(R[SL], W[SL], (int)((ST) & 1));                      \
    __builtin_amdgcn_sched_barrier(0);                        \
    issue(R[SL], W[SL], I[SL]);       /* stage ST + Depth */  \
    __builtin_amdgcn_sched_barrier(0);                        \
    load_idx(I[SL], (ST) + 2 * kHfDepth);                     \
    __builtin_amdgcn_sched_barrier(0);                        \
    lds_barrier();
    // whole groups of four steps, NO exit inside a group: an early exit is lowered to a common latch block that merges
    // the different "what is in flight" states of the steps, and the backend then waits for everything at the loop head
    // (vmcnt(0) in the ISA).  The last group runs past the row's end on clamped positions: up to three stages of loads
    // that hit the lines just read, parked and never consumed (the consumer keeps the same barrier count).
    for (int64_t s = 0; s < nstp; s += 4) {
      GGL_HF_STEP(0, s)
      GGL_HF_STEP(1, s + 1)
      GGL_HF_STEP(2, s + 2)
      GGL_HF_STEP(3, s + 3)
    }
#undef GGL_HF_STEP
    return;
  }
  // ---- consumer: one column per lane, the elements of a stage in order ---------------------------------------------
  if constexpr (F64) {
    double acc = 0.0;
    const int ndbl = ncol >> 1;          // (an even number of words: whole doubles)
    for (int64_t s = 0; s < nstp; ++s) {
      lds_barrier();
      const int b = (int)(s & 1);
      if (lane < ndbl && s < nst) {
        const int cnt = (int)((len - s * kHfStage) < kHfStage ? (len - s * kHfStage) : kHfStage);
        int e = 0;
        for (; e + 8 <= cnt; e += 8) {
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const double *>(&buf[b][e + q][2 * lane]);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc = __dadd_rn(acc, v[q]);
        }
        for (; e < cnt; ++e) acc = __dadd_rn(acc, *reinterpret_cast<const double *>(&buf[b][e][2 * lane]));
      }
    }
    if (lane < ndbl) reinterpret_cast<double *>(a.partial)[j * (a.K >> 1) + (c0 >> 1) + lane] = acc;
    return;
  }
  float acc = 0.0f;
  for (int64_t s = 0; s < nstp; ++s) {
    lds_barrier();                       // stage s is parked (the producers go on to park stage s + 1 in the other half)
    const int b = (int)(s & 1);
    if (lane < ncol && s < nst) {
      const int cnt = (int)((len - s * kHfStage) < kHfStage ? (len - s * kHfStage) : kHfStage);
      int e = 0;
      if (PIPE && cnt == kHfStage) {
        // a full stage, software-pipelined (round 5): the NEXT eight ds_read_b32 are in flight while the current eight
        // dependent adds retire — LDS returns in order, so the wait before a group of adds is lgkmcnt(8), not (0).  The
        // add chain (4 cycles per dependent v_add_f32) is then the only thing on the critical path inside a stage; the
        // first group's LDS latency is exposed once per stage.  Same adds, same order, same bits.
        float va[8], vb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = buf[b][q][lane];
#pragma unroll
        for (int g = 0; g < kHfStage; g += 16) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) vb[q] = buf[b][g + 8 + q][lane];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc = __fadd_rn(acc, va[q]);
          __builtin_amdgcn_sched_barrier(0);
          if (g + 16 < kHfStage) {
#pragma unroll
            for (int q = 0; q < 8; ++q) va[q] = buf[b][g + 16 + q][lane];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc = __fadd_rn(acc, vb[q]);
        }
        e = kHfStage;
      }
      for (; e + 8 <= cnt; e += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = buf[b][e + q][lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __fadd_rn(acc, v[q]);
      }
      for (; e < cnt; ++e) acc = __fadd_rn(acc, buf[b][e][lane]);
    }
  }
  if (lane < ncol) a.partial[j * a.K + c0 + lane] = acc;
}

// ---- the side streams the hub launch runs on (two per device, created on first use) ----------------------------------
// One for callers whose stream is being recorded into a hipGraph, one for eager callers: an eager launch must never land on a
// stream that another thread's capture has pulled in through its fork event (it would be recorded into that graph, or
// fail).  Every call takes its OWN join event from a small ring (round 4 shared one per device: a later record from a
// capturing stream could leave an eager stream waiting on a captured event); the ring is sized far above the number of
// hub launches that can be un-joined at once (one per host thread inside the library).
constexpr int kHubJoinRing = 32;
struct HubSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join[kHubJoinRing] = {};
  int next = 0;
  bool ok = false;
  // two host threads (the caller's and an autograd worker) may launch through here: the fork-record .. launch ..
  // join-record sequence of one call must not interleave with another's (fork is shared, the FIFO order is the contract)
  std::mutex mu;
};
static HubSide *hub_side(bool capturing, bool high = false) {
  static HubSide sides[16][4];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  HubSide &s = sides[dev][(capturing ? 1 : 0) + (high ? 2 : 0)];
  static std::mutex create_mu;
  std::lock_guard<std::mutex> g(create_mu);
  if (!s.ok) {
    // (a high-priority queue for the hub walk was measured and changes nothing: products step 74.99 vs 75.05 ms, bspmm
    //  16 x 16 forward 17.39 vs 17.56 — profiles/r4_negative_results.txt)
    // (high: the queue with the device's greatest priority — option hub_priority, for the ONE hub launch of a column-blocked
    //  aggregate, whose workgroups are handed out over the whole aggregate in competition with the row walks')
    int least = 0, greatest = 0;
    bool good = true;
    if (high) {
      good = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
             hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, greatest) == hipSuccess;
    } else {
      good = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
    }
    good = good && hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; good && i < kHubJoinRing; ++i) good = hipEventCreateWithFlags(&s.join[i], hipEventDisableTiming) == hipSuccess;
    if (!good) {      // nothing half-made stays behind (and the next call starts from scratch)
      for (int i = 0; i < kHubJoinRing; ++i)
        if (s.join[i]) { (void)hipEventDestroy(s.join[i]); s.join[i] = nullptr; }
      if (s.fork) { (void)hipEventDestroy(s.fork); s.fork = nullptr; }
      if (s.stream) { (void)hipStreamDestroy(s.stream); s.stream = nullptr; }
      (void)hipGetLastError();
      return nullptr;
    }
    s.ok = true;
  }
  return &s;
}
static bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

// Launch the exact-order walk of the plan's long rows.  With `beside` the launch goes to one of the library's side
// streams, forked from `stream` here; the caller launches its kernel(s) over the other rows on `stream` and then calls
// hub_f32_join(stream, *forked) before anything reads `partial`.  *forked = 0: not forked; otherwise the join token.
int hub_f32_launch(const HubF32Args &a, hipStream_t stream, bool beside, int *forked) {
  *forked = 0;
  if (a.n_long <= 0 || a.K <= 0) return GGL_OK;
  hipStream_t s = stream;
  const bool capturing = beside && stream_is_capturing(stream);
  // the greatest-priority queue (option hub_priority, off by default: common.hpp) only for big eager launches: a recorded
  // arxiv-sized step (0.3 ms aggregates, replayed from a hipGraph) ran 3.1 -> 4.3 ms with it
  const bool high = beside && options().hub_priority != 0 && !capturing &&
                    a.avg_long_len * a.n_long >= ((int64_t)1 << 22);
  HubSide *side = beside ? hub_side(capturing, high) : nullptr;
  std::unique_lock<std::mutex> lock;
  int token = 0;
  if (side != nullptr) {
    lock = std::unique_lock<std::mutex>(side->mu);
    GGL_HIP_CHECK(hipEventRecord(side->fork, stream));
    GGL_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
    s = side->stream;
    token = 1 + side->next + (capturing ? kHubJoinRing : 0) + (high ? 2 * kHubJoinRing : 0);
    side->next = (side->next + 1) % kHubJoinRing;
  }
  const bool narrow = a.K <= 16;                       // 16-column slabs, 4 lanes per element
  const int64_t slabs = ceil_div(a.K, (int64_t)(narrow ? 16 : 64));
  const int64_t grid = a.n_long * slabs;
  const bool vec4 = a.K % 4 == 0 && a.x_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15u) == 0;
  const bool seg = a.col == nullptr, has_w = !seg && a.w != nullptr;
  const bool w_perm = has_w && !a.w_by_pos && a.perm != nullptr;
  const bool heads = has_w && a.C > 0, wpc = heads && a.C % 4 != 0;
  // (a weight per column costs 48 more registers per lane at PER = 4: those rare shapes — heads whose channel count is
  //  not a multiple of 4 — take the light configuration, which does not spill)
  const bool heavy = a.avg_long_len >= kHfHeavyFrom && !(a.C > 0 && a.C % 4 != 0 && a.w != nullptr);
#define GGL_HF2(M, W, V)                                                                          \
  do {                                                                                             \
    if (narrow) GGL_LAUNCH((hub_rows_f32_kernel<M, V, W, 2, 2>), grid, kHfBlock, s, a);            \
    else if (heavy && options().hub_pipe == 0)  /* A/B: round 4's consumer (read 8, add 8, no overlap) */ \
      GGL_LAUNCH((hub_rows_f32_kernel<M, V, W, 4, 4, false, false>), grid, kHfBlock, s, a);        \
    else if (heavy) GGL_LAUNCH((hub_rows_f32_kernel<M, V, W, 4, 4>), grid, kHfBlock, s, a);        \
    else GGL_LAUNCH((hub_rows_f32_kernel<M, V, W, 2, 4>), grid, kHfBlock, s, a);                   \
  } while (0)
#define GGL_HF(M, W)                                   \
  do {                                                 \
    if (vec4) GGL_HF2(M, W, true);                     \
    else GGL_HF2(M, W, false);                         \
  } while (0)
  if (seg && a.f64) {   // (light stages: one instantiation per index mode and width class is enough for this rare dtype)
#define GGL_HF64(M)                                                                                       \
  do {                                                                                                     \
    if (narrow) {                                                                                          \
      if (vec4) GGL_LAUNCH((hub_rows_f32_kernel<M, true, false, 2, 2, true>), grid, kHfBlock, s, a);       \
      else GGL_LAUNCH((hub_rows_f32_kernel<M, false, false, 2, 2, true>), grid, kHfBlock, s, a);           \
    } else if (vec4) {                                                                                     \
      GGL_LAUNCH((hub_rows_f32_kernel<M, true, false, 2, 4, true>), grid, kHfBlock, s, a);                 \
    } else {                                                                                               \
      GGL_LAUNCH((hub_rows_f32_kernel<M, false, false, 2, 4, true>), grid, kHfBlock, s, a);                \
    }                                                                                                      \
  } while (0)
    if (a.perm) GGL_HF64(HUB_SEG_PERM); else GGL_HF64(HUB_SEG);
#undef GGL_HF64
  } else if (seg) {
    if (a.perm) GGL_HF(HUB_SEG_PERM, false); else GGL_HF(HUB_SEG, false);
  } else if (!has_w) {
    GGL_HF(HUB_SPMM, false);
  } else if (!heads) {
    if (w_perm) GGL_HF(HUB_SPMM_WP, false); else GGL_HF(HUB_SPMM_W, false);
  } else if (!wpc) {
    if (w_perm) GGL_HF(HUB_BSPMM_WP, false); else GGL_HF(HUB_BSPMM_W, false);
  } else {
    if (w_perm) GGL_HF(HUB_BSPMM_WP, true); else GGL_HF(HUB_BSPMM_W, true);
  }
#undef GGL_HF
#undef GGL_HF2
  GGL_LAUNCH_CHECK();
  if (token) {
    GGL_HIP_CHECK(hipEventRecord(side->join[(token - 1) % kHubJoinRing], side->stream));
    *forked = token;
  }
  return GGL_OK;
}

int hub_f32_join(hipStream_t stream, int token) {
  if (token <= 0) return GGL_OK;
  const int which = (token - 1) / kHubJoinRing;          // bit 0: capturing caller, bit 1: high-priority queue
  HubSide *side = hub_side((which & 1) != 0, (which & 2) != 0);
  GGL_REQUIRE(side != nullptr, GGL_EHIP, "hub side stream is gone");
  GGL_HIP_CHECK(hipStreamWaitEvent(stream, side->join[(token - 1) % kHubJoinRing], 0));
  return GGL_OK;
}

}  // namespace ggl
