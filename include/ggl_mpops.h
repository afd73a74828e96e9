/*
 * include/ggl_mpops.h — C ABI of libggl_mpops_hip.so: the MI355X (gfx950) message-passing backend
 * that replaces GammaGL's gammagl/mpops/torch_ext (`_torch_ext`) for the torch backend.
 *
 * Boundary.  The reference binds its native ops to Python through a pybind11 module with seven
 * free functions (gammagl/mpops/torch_ext/src/operators.cpp:51-59):
 *     c_segment_sum / c_segment_mean / c_segment_max (Tensor x, Tensor index, int64 N) -> Tensor
 *     c_spmm_sum / c_spmm_mean / c_spmm_max / c_bspmm_sum (Tensor index, Tensor weight, Tensor x)
 * each a torch::autograd::Function (include/<op>.h, src/<op>.cpp) that dispatches to a CPU loop
 * (cpu/<op>_cpu.cpp) or an atomic CUDA kernel (cuda/<op>_cuda.cu).  This header is what an FFI for that path
 * binds instead: plain device pointers, sizes and a HIP stream — no torch types.  Every entry
 * point cites the reference function it supersedes.  The ctypes stub a GammaGL maintainer would
 * add is in INTEGRATION.md; gammagl_amd/mpops.py is that stub, complete.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers on the current HIP device unless named *_host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every compute entry
 *     point is asynchronous and stream-ordered and performs NO host synchronisation; the
 *     ggl_plan_* entry points are synchronous (they return host-side facts about the graph) and are
 *     meant to run once per edge list, not once per step.
 *   - Return value: GGL_OK or a negative GGL_E* code; ggl_last_error() gives a message.
 *   - Feature rows are contiguous, row-major: x[e*K + k].  K = product of trailing dims
 *     (segment_sum_cpu.cpp:44: K = x.numel() / x.size(0)).
 *   - No atomics anywhere on the reduction path: every output row is produced by exactly one
 *     wavefront group that walks the row's edges in ascending ORIGINAL edge order, which makes the
 *     results run-to-run deterministic and, for rows shorter than the long-row threshold,
 *     bit-identical to the reference's serial CPU loops (same order of the same rounded adds).
 */
#ifndef GGL_MPOPS_H
#define GGL_MPOPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 3): + ggl_bspmm_grad_w_sorted[_scratch_bytes]; v4's number had not been raised for the symbols added
 * late in round 2 (ggl_sample_hop, ggl_block_transpose, ggl_gat_sh_*, ggl_segment_hub16*, ggl_spmm_col_blocks) */
/* 6 (round 4): + ggl_calib_stream (bench.py's achievable-rate yardstick) */
/* 7 (round 4): ggl_segplan_t GREW by `long_order` and `max_len` (appended; hubf32.hip's serial hub walk); + ggl_spmm_max_bwd32;
 *   ggl_policy_gradw_sorted takes (H, C).  Because the struct grew, a caller compiled against ABI <= 6 would hand the
 *   library a SHORTER struct than it reads: every caller MUST check ggl_abi_version() == GGL_ABI_VERSION before it passes
 *   a ggl_segplan_t (gammagl_amd/_lib.py bind(), ggl_torch.cpp api_for() both refuse a mismatching library). */
/* 8 (round 5): + ggl_invert_perm, ggl_spmm_max_mask[_bytes], ggl_spmm_max_bwd_mask (gspmm max backward through a 1-bit
 *   winner mask), ggl_spmm_max_mask_words; ggl_spmm_max_bwd_mask takes mask_pos; ggl_gat_fast_bwd's plan->partial holds four
 *   doubles per chunk and head; ggl_segplan_t.xcd_run_rows < 0 is a hint (see the field); + ggl_sample_hop_ex; options
 *   hub_one_launch, hub_priority, hub_pipe, maxbwd_mask*, gat_sh_waves, hop_fused_scans.  No struct change. */
/* 9 (round 6): + ggl_policy_maxbwd_form (the gspmm-max backward's form gated on the winner mask's footprint; option
 *   maxbwd_mask_kmax), ggl_gat_sh_bwd's forward-plan partial holds four doubles per chunk and head, + ggl_spmm_col_blocks_plan (128-column blocks for plans whose node order carries locality), option gat_sh_pk.
 *   No struct change. */
#define GGL_ABI_VERSION 9

/* dtype codes (AT_DISPATCH_ALL_TYPES_AND2(Half, BFloat16), segment_sum_cpu.cpp:32-33) */
enum {
  GGL_U8 = 0, GGL_I8 = 1, GGL_I16 = 2, GGL_I32 = 3, GGL_I64 = 4,
  GGL_F16 = 5, GGL_BF16 = 6, GGL_F32 = 7, GGL_F64 = 8
};

enum {
  GGL_OK = 0,
  GGL_EINVAL = -1,   /* bad argument (TORCH_CHECK in the reference -> RuntimeError) */
  GGL_EINDEX = -2,   /* id out of range (TORCH_CHECK_INDEX -> IndexError, segment_max_cpu.cpp:50) */
  GGL_EDTYPE = -3,   /* unsupported dtype ("expected scalar type Float", spmm_sum_cpu.cpp:22) */
  GGL_EHIP = -4,     /* HIP runtime error */
  GGL_EWORKSPACE = -5 /* workspace too small */
};

int ggl_abi_version(void);
const char *ggl_last_error(void);          /* thread-local message of the last failing call */
/* compute-unit count, wavefront size and gfx arch string of the current device (host query) */
int ggl_device_info(int *cus_host, int *wave_host, char *arch_host, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Segment plan: a destination-sorted (CSR-like) view of an unsorted id vector.
 * The reference walks `index[e]` for e = 0..E-1 and scatters (segment_sum_cpu.cpp:47-56;
 * CUDA: one atomicAdd per (e,k), segment_sum_cuda.cu:19-31).  Here the ids are stably sorted once
 * per edge list (rocPRIM LSD radix sort on (id, e) pairs), giving
 *   perm[p]   : original element index of the p-th element in sorted order (ascending e inside
 *               a segment), or no perm at all when the ids already arrive sorted;
 *   rowptr[s] : first sorted position of segment s (int64, N+1 entries).
 * Rows longer than `chunk` are listed in long_rows and are reduced chunk-by-chunk into a partial
 * buffer and then combined in chunk order (still deterministic).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ggl_segplan {
  const int64_t *rowptr;    /* [N+1] */
  const int32_t *perm;      /* [E] or NULL when the ids were already sorted */
  const int32_t *long_rows; /* [n_long] row ids with more than `chunk` elements, or NULL */
  const int64_t *chunk_ptr; /* [n_long+1] exclusive prefix of per-long-row chunk counts, or NULL */
  int64_t n_long;
  int64_t n_chunks;         /* chunk_ptr[n_long] */
  int64_t chunk;            /* elements per chunk == long-row threshold (> 0) */
  void *partial;            /* workspace: ggl_partial_bytes(...) bytes, or NULL when n_long == 0 */
  int64_t N;                /* number of segments (output rows) */
  int64_t E;                /* number of elements (edges) */
  const int32_t *row_order; /* [N] rows sorted by length (longest first) or NULL: the order in which
                               row slots are handed to wavefronts — scheduling only, results identical */
  int64_t xcd_run_rows;     /* > 0: the node order carries locality — hand each XCD runs of this many consecutive row
                               slots (its private L2 then serves one neighbourhood instead of 1/8 of every one);
                               0: the library default (round-robin); < 0 (ABI 8): no runs, but the plan's long rows LEAD the id
                               range (a degree-sorted node order) — a column-blocked aggregate then walks its hub rows once
                               over the full width (option hub_one_launch = 2).  Scheduling only (ABI 5). */
  const int32_t *long_order; /* [n_long] positions in long_rows by descending row length, or NULL: the order in which the
                               serial hub walk (hubf32.hip) starts its rows — longest first, so that the one add chain
                               nobody can shorten runs under everything else.  Scheduling only (ABI 6). */
  int64_t max_len;          /* elements of the longest row (ggl_plan_build's *max_len_host), or 0 = unknown.  The serial hub
                               walk adds a row's elements one after the other (~3.5 ns each): a plan whose longest row
                               exceeds the option `exact_long_max` (2^21 elements; a star graph's centre) keeps the
                               chunked walk — within rounding of the reference instead of its bits — rather than wait
                               for one add chain (ABI 7). */
} ggl_segplan_t;

/* bytes of workspace ggl_plan_build needs */
size_t ggl_plan_workspace_bytes(int64_t E, int64_t N);

/* Build perm/rowptr from ids[E] (int64 on device, the reference's index dtype:
 * segment_sum_cpu.cpp:36 data_ptr<int64_t>).  SYNCHRONOUS: one host read for E <= 2^22 (flags, sort, row
 * pointer and longest row are queued behind it: a fresh edge list per mini-batch costs ~0.1 ms), two above (the
 * first decides whether an already sorted 10^8-element list skips its sort).  perm may be written even when the
 * input turns out sorted (then *is_sorted_host = 1 and the caller may drop it).
 * Errors: GGL_EINDEX if any id < 0 or >= N (the reference: IndexError for max,
 * segment_max_cpu.cpp:50; silent out-of-bounds write for sum/mean). */
int ggl_plan_build(const int64_t *ids, int64_t E, int64_t N, int32_t *perm, int64_t *rowptr,
                   void *workspace, size_t workspace_bytes, void *stream, int32_t *is_sorted_host,
                   int64_t *max_len_host);

/* Long-row bookkeeping (SYNCHRONOUS): count, then fill long_rows[n_long] (ascending row id) and
 * chunk_ptr[n_long+1].  Both need ggl_plan_long_workspace_bytes(N) bytes of scratch. */
size_t ggl_plan_long_workspace_bytes(int64_t N);
int ggl_plan_long_count(const int64_t *rowptr, int64_t N, int64_t chunk, void *workspace,
                        size_t workspace_bytes, void *stream, int64_t *n_long_host,
                        int64_t *n_chunks_host);
int ggl_plan_long_fill(const int64_t *rowptr, int64_t N, int64_t chunk, int64_t n_long,
                       int32_t *long_rows, int64_t *chunk_ptr, void *workspace,
                       size_t workspace_bytes, void *stream);
/* bytes of `partial` for n_chunks chunks of K features of dtype (value + int64 arg for max) */
size_t ggl_partial_bytes(int dtype, int64_t n_chunks, int64_t K, int with_arg);

/* p[0..n) = v */
int ggl_fill_i64(int64_t *p, int64_t n, int64_t v, void *stream);

/* out_i32[p] = (int32) src_i64[perm ? perm[p] : p]   — e.g. the CSR column array from edge_index[0] */
int ggl_gather_i64_to_i32(const int64_t *src, const int32_t *perm, int64_t E, int32_t *out,
                          void *stream);
/* out[p, :] = src[perm[p], :] for rows of H floats (edge weights into sorted order) */
int ggl_gather_rows_f32(const float *src, const int32_t *perm, int64_t E, int64_t H, float *out,
                        void *stream);
/* out[i, 0:K] = src[idx[i], 0:K], row strides src_ld / out_ld (elements, >= K): the send buffer of one
 * feature-column block of the halo exchange gathered straight from the activation matrix (no reference
 * counterpart: SURVEY.md §8e; replaces h[:, c0:c1].index_select(0, idx).contiguous()) */
int ggl_gather_rows_f32_ex(const float *src, int64_t src_ld, const int64_t *idx, int64_t n, int64_t K,
                           float *out, int64_t out_ld, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Format conversion: device-side ind2ptr / ptr2ind (gammagl/ops/sparse,
 * cpu/convert.cpp:58-128, cuda/convert.cu:41-104) and sort_edge_index (utils/sort_edge_index.py:30-44).
 *   ggl_ind2ptr   : ptr[M+1] = exclusive prefix of the histogram of ind[E] (int64; SYNCHRONOUS:
 *                   validates 0 <= ind < M like ggl_plan_build)
 *   ggl_ptr2ind   : ind[p] = r with ptr[r] <= p < ptr[r+1]
 *   ggl_sort_edges: perm[E] = stable argsort of major[i] * N + minor[i]
 * ---------------------------------------------------------------------------------------------- */
size_t ggl_ind2ptr_workspace_bytes(int64_t E, int64_t M);
int ggl_ind2ptr(const int64_t *ind, int64_t E, int64_t M, int64_t *ptr, void *workspace,
                size_t workspace_bytes, void *stream);
int ggl_ptr2ind(const int64_t *ptr, int64_t M, int64_t E, int64_t *ind, void *stream);
size_t ggl_sort_edges_workspace_bytes(int64_t E, int64_t N);
int ggl_sort_edges(const int64_t *major, const int64_t *minor, int64_t E, int64_t N, int32_t *perm,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Segment reductions — supersede segment_{sum,mean,max}_{cpu,cuda}_forward
 * (cpu/segment_sum_cpu.cpp:11-60, cpu/segment_mean_cpu.cpp:11-80, cpu/segment_max_cpu.cpp:11-69;
 *  cuda/segment_sum_cuda.cu:33-116, cuda/segment_mean_cuda.cu, cuda/segment_max_cuda.cu).
 * x [E,K] of `dtype`, out [N,K] (every element written).  Semantics are those of the CPU
 * extension: accumulate in the storage dtype; mean = sum / count with count held in the storage
 * dtype and applied only where count > 1 (integer dtypes: truncating division); max pre-fills
 * with numeric_limits<T>::lowest(), strict `<` so the smallest e wins ties and NaN never wins;
 * if E*K == 0 max returns zeros.  arg [N,K] int64 = winning element index, `arg_fill` for empty
 * segments (the caller passes E; the reference's N aliases real rows — see DESIGN.md).
 * ---------------------------------------------------------------------------------------------- */
int ggl_segment_sum(int dtype, const void *x, const ggl_segplan_t *plan, int64_t K, void *out,
                    void *stream);
int ggl_segment_mean(int dtype, const void *x, const ggl_segplan_t *plan, int64_t K, void *out,
                     void *stream);
/* Wide f32 SpMM-sum / mean launches are cut into column blocks (64 wide; csrc/reduce.hip launch_f32_cols, options
 * "col_block" / "col_block_min_edges" / "col_block_min_degree"): the number of kernel launches one call over E edges,
 * K columns and N output rows makes. */
int64_t ggl_spmm_col_blocks(int64_t E, int64_t K, int64_t N);
/* ... for a plan: as above, blocks twice as wide where plan->xcd_run_rows > 0 (a node order with locality; ABI 9) */
int64_t ggl_spmm_col_blocks_plan(const ggl_segplan_t *plan, int64_t K);

/* f16 / bf16 sums accumulate in the storage type (segment_sum_cpu.cpp:47-56), so their hub rows cannot be chunked:
 * ggl_segment_hub16 reduces the plan's LONG rows (plan->long_rows) in the reference's serial order with a workgroup
 * per (row, 64-column slab) that prefetches through LDS (GPU build only: ggl_segment_hub16_supported; rows of a
 * multiple of 8 columns move as 16-byte pieces, other widths element by element).  Pair it with ggl_segment_{sum,mean} on the same plan with the long-row table withheld
 * (n_long = 0, chunk kept): that launch skips rows longer than chunk. */
int ggl_segment_hub16_supported(int dtype, int64_t K, const void *x, const void *out);
int ggl_segment_hub16(int dtype, int mean, const void *x, const ggl_segplan_t *plan, int64_t K, void *out,
                      void *stream);
int ggl_segment_max(int dtype, const void *x, const ggl_segplan_t *plan, int64_t K, void *out,
                    int64_t *arg, int64_t arg_fill, void *stream);

/* Backward passes — supersede SegmentSum/Mean/Max::backward (src/segment_sum.cpp:43-54,
 * src/segment_mean.cpp:44-63, src/segment_max.cpp:48-61).  f16/bf16/f32/f64 gradients.
 *   sum : gin[e,:]  = gout[ids[e],:]
 *   mean: gin[e,:]  = gout[ids[e],:] / count[ids[e]]       (count from rowptr)
 *   max : gin = 0;  gin[arg[s,k],k] = gout[s,k] for arg < E (empty segments contribute nothing) */
int ggl_segment_sum_bwd(int dtype, const void *gout, const int64_t *ids, int64_t E, int64_t K,
                        void *gin, void *stream);
int ggl_segment_mean_bwd(int dtype, const void *gout, const int64_t *ids, const int64_t *rowptr,
                         int64_t E, int64_t K, void *gin, void *stream);
int ggl_segment_max_bwd(int dtype, const void *gout, const int64_t *arg, int64_t E, int64_t N,
                        int64_t K, void *gin, void *stream);

/* ------------------------------------------------------------------------------------------------
 * gspmm — CSR x dense SpMM, f32; supersedes spmm_{sum,mean,max}_cpu_{forward,backward}
 * (cpu/spmm_sum_cpu.cpp:5-80, cpu/spmm_mean_cpu.cpp:5-105, cpu/spmm_max_cpu.cpp:5-99) and
 * spmm_sum_cuda_{forward,backward} (cuda/spmm_sum_cuda.cu:15-87).
 * plan  : rows = destination nodes (N_out = plan->N), sorted positions p = rowptr[i]..rowptr[i+1]
 * col   : [E] int32, source node of sorted position p
 * w     : [E] f32 edge weights or NULL (= all ones).  w_by_pos != 0: indexed by sorted position p;
 *         w_by_pos == 0: indexed by original edge id (through plan->perm when present).
 * x     : [N_in, K] f32;  out : [N_out, K] f32
 *   sum : out[i,:] = sum_p  w[p] * x[col[p],:]          (rounded multiply, then rounded add)
 *   mean: sum / (number of edges into i), only where that count > 0   (spmm_mean_cpu.cpp:51-58)
 *   max : out pre-filled with -FLT_MAX; arg [N_out,K] int64 = SOURCE NODE id of the winner
 *         (spmm_max_cpu.cpp:47), 0 where the row is empty.
 * The backward of `sum` w.r.t. x is the same call on the transposed plan (rows = source nodes,
 * col = destination nodes), which is exactly spmm_sum_cpu_backward's gx[src] += w[e] * g[dst].
 * ---------------------------------------------------------------------------------------------- */
/* Strided / accumulating forms of ggl_spmm_sum and ggl_segment_sum: x rows x_ld elements apart, out rows
 * out_ld apart (0 = K; >= K otherwise), and with accumulate != 0 the row sums are ADDED to what out holds
 * (the previous value first in the summation order).  They let the multi-GPU layer aggregate a column block
 * of a wider matrix in place and add the halo-source edges onto the local-source result without a
 * temporary and an extra pass. */
int ggl_spmm_sum_ex(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                    const float *x, int64_t x_ld, int64_t K, float *out, int64_t out_ld, int accumulate,
                    void *stream);
int ggl_segment_sum_ex(int dtype, const void *x, int64_t x_ld, const ggl_segplan_t *plan, int64_t K,
                       void *out, int64_t out_ld, int accumulate, void *stream);
int ggl_spmm_sum(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                 const float *x, int64_t K, float *out, void *stream);
int ggl_spmm_mean(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                  const float *x, int64_t K, float *out, void *stream);
int ggl_spmm_max(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                 const float *x, int64_t K, float *out, int64_t *argsrc, void *stream);
/* planT: rows = source nodes; colT[p] = destination node; fwd_rowptr = forward plan's rowptr
 *   mean bwd: gx[j,:] = sum_p (g[colT[p],:] / count[colT[p]]) * w[p]     (spmm_mean_cpu.cpp:95-101)
 *   max  bwd: gx[j,k] = sum_p [argsrc[colT[p],k] == j] w[p] * g[colT[p],k] (spmm_max_cpu.cpp:88-93) */
int ggl_spmm_mean_bwd(const ggl_segplan_t *planT, const int32_t *colT, const float *w, int w_by_pos,
                      const float *g, const int64_t *fwd_rowptr, int64_t K, float *gx,
                      void *stream);
int ggl_spmm_max_bwd(const ggl_segplan_t *planT, const int32_t *colT, const float *w, int w_by_pos,
                     const float *g, const int64_t *argsrc, int64_t K, float *gx, void *stream);
/* the same walk with the witnesses in a compact int32 copy of argsrc (node ids index int32 arrays everywhere in this
 * library): the lookup is per edge AND column, 8 of the walk's 12 bytes per element with int64 witnesses.  NOT yet the
 * hosts' default: unmeasured on the GPU (the Engine takes it with the option `maxbwd_arg32`; results identical). */
int ggl_spmm_max_bwd32(const ggl_segplan_t *planT, const int32_t *colT, const float *w, int w_by_pos,
                       const float *g, const int32_t *argsrc32, int64_t K, float *gx, void *stream);
/* The max backward through a WINNER MASK (round 5; same sums in the same order as ggl_spmm_max_bwd, spmm_max_cpu.cpp:88-93):
 *   ggl_spmm_max_mask  walks the FORWARD plan (rows = destinations, where argsrc's row is wave-uniform) and writes one
 *     record per edge, bit k = [argsrc[dst, k] == colF[p]] for the edge at forward position p:
 *       tpos == NULL (the hosts' default): records in FORWARD position order, ggl_spmm_max_mask_words(K, 1) words each
 *         (ceil(K/32) rounded up to 1, 2, 4 or a multiple of 8), written as coalesced 32-record blocks; the backward
 *         reads record posT[t] for transposed position t (pass mask_pos = posT, the hosts' GraphPlan.posT);
 *       tpos != NULL: records SCATTERED to transposed position tpos[p] (ggl_invert_perm of posT), ceil(K/32) words
 *         each; the backward streams them in its own order (mask_pos = NULL).  Measured slower (the scatter), kept as A/B.
 *     bit k of a record: word k/32, bit k%32.  mask: ggl_spmm_max_mask_bytes(E, K) bytes (room for either form), 16-byte
 *     aligned, every record fully overwritten;
 *   ggl_spmm_max_bwd_mask  is the transposed walk reading K/8 mask bytes per edge instead of 8K witness bytes. */
int64_t ggl_spmm_max_mask_words(int64_t K, int forward_order);
size_t ggl_spmm_max_mask_bytes(int64_t E, int64_t K);
int ggl_spmm_max_mask(const ggl_segplan_t *planF, const int32_t *colF, const int32_t *tpos, const int64_t *argsrc,
                      int64_t K, uint32_t *mask, void *stream);
int ggl_spmm_max_bwd_mask(const ggl_segplan_t *planT, const int32_t *colT, const float *w, int w_by_pos,
                          const float *g, const uint32_t *mask, const int32_t *mask_pos, int64_t K, float *gx,
                          void *stream);
/* inv[perm[i]] = i for a permutation of [0, n), n < 2^31 */
int ggl_invert_perm(const int32_t *perm, int64_t n, int32_t *inv, void *stream);

/* ------------------------------------------------------------------------------------------------
 * bspmm — multi-head SpMM, f32; supersedes bspmm_sum_cpu_{forward,backward}
 * (cpu/bspmm_sum_cpu.cpp:7-56, 58-113).  x [N_in,H,C], w [E,H], out [N_out,H,C]:
 *   out[i,h,c] = sum_p w[p,h] * x[col[p],h,c]
 * backward: gx = the same call on the transposed plan with g in place of x;
 *           gw[e,h] = sum_c x[src[e],h,c] * g[dst[e],h,c]  (edge-parallel, original edge order).
 * ---------------------------------------------------------------------------------------------- */
int ggl_bspmm_sum(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                  const float *x, int64_t H, int64_t C, float *out, void *stream);
int ggl_bspmm_grad_w(const int64_t *index /* [2,E] int64 */, const float *x, const float *g,
                     int64_t E, int64_t H, int64_t C, float *gw, void *stream);
/* The same weight gradient (bspmm_sum_cpu.cpp:95-107: serial over c, rounded multiply then rounded add — bit for
 * bit) computed along the destination-sorted FORWARD plan: col[p] / rowidx[p] = source / destination node of sorted
 * position p, the result goes to gw[perm[p], h] (the caller's edge order).  Strips are staged through LDS with
 * coalesced 16-byte loads (csrc/edgedot.hip); `scratch` = ggl_bspmm_grad_w_sorted_scratch_bytes(...) bytes (0 -> may be
 * NULL): wide heads then run as launches over 64-column blocks that carry the running dot in it. */
size_t ggl_bspmm_grad_w_sorted_scratch_bytes(int64_t E, int64_t N, int64_t H, int64_t C);
int ggl_bspmm_grad_w_sorted(const ggl_segplan_t *plan, const int32_t *col, const int32_t *rowidx, const float *x,
                            const float *g, int64_t H, int64_t C, float *gw, float *scratch, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Column sums of a row-major [N,K] f32 matrix: out[k] = sum_r g[r,k] — the gradient of the
 * "+ bias" that follows every aggregate (gcn_conv.py:105-106, sage_conv.py:102-103).  Two
 * deterministic stages (no atomics); needs ggl_colsum_workspace_bytes(N, K) bytes of scratch.
 * ---------------------------------------------------------------------------------------------- */
size_t ggl_colsum_workspace_bytes(int64_t N, int64_t K);
int ggl_colsum_f32(const float *g, int64_t N, int64_t K, float *out, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused epilogue of an aggregate: "+ bias" (gcn_conv.py:105-106), ReLU and
 * dropout (models/gcn.py:55-59) in one pass each way.
 *   fwd: y = keep * relu(a + bias) / (1 - p_drop); keep ~ Bernoulli(1 - p_drop) from Philox4x32-10 keyed
 *        on rng_state = {seed, offset} (device int64[2]; offset is advanced on the stream after the
 *        launch, so a captured graph draws a new mask per replay).  bias [K] or NULL; p_drop = 0: no RNG.
 *        The word of element (r, k): Philox(counter = r * (K / v) + k / v)[k % v], v = 4 if K % 4 == 0 else 1.
 *   bwd: ga = keep * [relu ? y > 0 : 1] * g / (1 - p_drop).  With ReLU the mask is read off y (y > 0: exact).
 *        Without ReLU `rng_used` = a copy of the {seed, offset} the forward READ (taken before it advanced)
 *        lets the dropout mask be redrawn exactly; rng_used == NULL: y != 0 stands in for it (exact except
 *        for kept activations that are exactly 0).  gbias[K] = column sums of ga (same pass; NULL to skip).
 * ---------------------------------------------------------------------------------------------- */
int ggl_bias_act_fwd(const float *a, const float *bias, int64_t N, int64_t K, int relu, float p_drop,
                     int64_t *rng_state, float *y, void *stream);
/* ggl_spmm_sum with that forward epilogue applied to every finished row before its only store:
 * out = dropout(relu(A x + bias)).  Identical (values and dropout mask) to ggl_spmm_sum followed by
 * ggl_bias_act_fwd on the same rng_state; the backward is ggl_bias_act_bwd(g, out) then the transposed
 * ggl_spmm_sum. */
int ggl_spmm_sum_bias_act(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                          const float *x, int64_t K, const float *bias, int relu, float p_drop,
                          int64_t *rng_state, float *out, void *stream);
size_t ggl_bias_act_bwd_workspace_bytes(int64_t N, int64_t K);
int ggl_bias_act_bwd(const float *g, const float *y, int64_t N, int64_t K, int relu, float p_drop,
                     const int64_t *rng_used, float *ga, float *gbias, void *workspace,
                     size_t workspace_bytes, void *stream);
/* General form of the fused epilogue (sage_conv.py:100-108, gcn_conv.py:105-106):
 *   out[i, 0:K] = dropout(relu(reduce_{p in row i} w * x[col[p], 0:K] (+ out[i, 0:K] if accumulate)
 *                              + add[i, 0:K] + bias[0:K]))
 * reduce = sum, or mean (divide by the row's edge count; not with accumulate).  x / out / add rows are
 * x_ld / out_ld / add_ld elements apart (0 = K): a column block [epi_col0, epi_col0 + K) of an epi_K-wide
 * row (epi_K = 0: K), bias / add pointing at the block's first column, and the dropout word of element
 * (i, epi_col0 + k) is the one the full-width launch draws.  accumulate: a second edge set (halo-source
 * edges of the multi-GPU path) is added onto the partial result already in `out`, epilogue included.
 * advance_rng != 0 steps rng_state afterwards (once per layer: on the last column block). */
int ggl_spmm_epi_ex(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                    const float *x, int64_t x_ld, int64_t K, float *out, int64_t out_ld, int accumulate,
                    int mean, const float *add, int64_t add_ld, const float *bias, int relu, float p_drop,
                    int64_t *rng_state, int64_t epi_K, int64_t epi_col0, int advance_rng, void *stream);
/* segment_sum / segment_mean of f32 messages x[E, K] with the same epilogue on each finished row (the
 * message() + aggregate() route of a sampled SAGEConv block): bit-identical to ggl_segment_{sum,mean}
 * followed by "+ add + bias, relu, dropout". */
int ggl_segment_epi(const float *x, const ggl_segplan_t *plan, int64_t K, int mean, const float *add,
                    int64_t add_ld, const float *bias, int relu, float p_drop, int64_t *rng_state, float *out,
                    void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused GAT edge-softmax + weighted aggregate: ONE kernel per direction.  Replaces the external
 * dgNN GATConvFuse that FusedGATConv calls (layers/conv/fusedgat_conv.py:70-71,121) and the
 * unfused chain in GATConv.forward (layers/conv/gat_conv.py:103-112 + utils/softmax.py:29-35):
 *   s[p,h]   = LeakyReLU_slope(el[col[p],h] + er[i,h])       for p in row i
 *   m[i,h]   = max_p s;  d[i,h] = sum_p exp(s - m)           (saved for backward)
 *   out[i,h,:] = sum_p exp(s[p,h]-m[i,h]) / (d[i,h] + 1e-16) * x[col[p],h,:]
 * el/er [N,H] f32 (source / destination attention terms), x [N_in,H,C] f32.
 * Rows longer than plan->chunk are reduced chunk-wise with chunk-local maxima and merged in chunk
 * order (online-softmax identity); short rows use the formula above verbatim.
 * Backward (edge-parallel, then source-major):
 *   ggl_gat_fused_bwd_dst: ONE walk of the forward plan (lane = head of a row / of a hub chunk):
 *       dot[i,h] = <g_i, out_i>; per (sorted position, head): alpha, de = alpha (<g_i, x_j> - dot)
 *       lrelu'(.) -> alpha[E,H], de[E,H] (forward sorted positions); ger[N,H] = row sums of de
 *   ggl_gat_fused_bwd_src: ONE walk of the transposed plan: gx[j,h,:] = sum_p alpha * g[dst,h,:] and
 *       gel[j,h] = sum_p de, reading alpha/de through posT (transposed position -> forward position);
 *       planT->partial = ggl_partial_bytes(GGL_F32, n_chunks, H*C + H, 0) bytes when it has long rows
 * ---------------------------------------------------------------------------------------------- */
/* p_drop > 0: attention dropout (gat_conv.py:104 `dropout(segment_softmax(.))`, GATConvFuse's last
 * argument): edge (sorted position p, head h) is kept when the 32-bit word mix(seed, offset, p * H + h)
 * (a counter-based multiply-xor mix with xxHash32's avalanche, gat.hip drop_word; ABI 4 — ABI 3 drew it from
 * Philox4x32-10, whose ten rounds per scattered position dominated the backward's source walks) is
 * >= p_drop * 2^32, and then weighs alpha / (1 - p_drop); the softmax itself runs over all edges.  rng_state = device int64 {seed, offset},
 * offset advanced after the launch; the backward needs the values the forward READ (rng_used). */
int ggl_gat_fused_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el,
                      const float *er, const float *x, float slope, int64_t H, int64_t C,
                      float p_drop, int64_t *rng_state, float *out, float *rowmax, float *rowden,
                      void *stream);
/* plan->partial for ggl_gat_fused_fwd when the plan has long rows (chunk-local softmax partials) */
size_t ggl_gat_partial_bytes(int64_t n_chunks, int64_t H, int64_t C);
/* alpha / de: two [E,H] f32 arrays, or ONE interleaved [E,H,2] buffer passed as (base, base + 1) — then
 * an edge's alpha and de share a 64-byte line for the source-side walk (same convention in
 * ggl_gat_fused_bwd_src).  rowidx, dot_ws: unused since ABI 3 (may be NULL); plan->partial = ggl_partial_bytes(GGL_F32, n_chunks,
 * H, 0) bytes when the plan has long rows (ger partials of the hub chunks) */
int ggl_gat_fused_bwd_dst(const ggl_segplan_t *plan, const int32_t *col, const int32_t *rowidx,
                          const float *el, const float *er, const float *x, const float *g,
                          const float *out, const float *rowmax, const float *rowden, float slope,
                          int64_t H, int64_t C, float p_drop, const int64_t *rng_used, float *alpha,
                          float *de, float *ger, float *dot_ws, void *stream);
int ggl_gat_fused_bwd_src(const ggl_segplan_t *planT, const int32_t *colT, const int32_t *posT,
                          const float *alpha, const float *de, const float *g, int64_t H,
                          int64_t C, float *gx, float *gel, void *stream);
/* Fast path of the same op for heads of C = 4, 8, 16, 32 or 64 channels with H * C <= 256
 * (ggl_gat_fast_supported; e.g. the Reddit GAT's 8 x 8): same walks with ~5x fewer vector-ALU instructions
 * (v_exp_f32, one rescale per 4-8 edges, 16-byte index loads, 32-bit panel offsets, FMA) and a backward that
 * RECOMPUTES alpha / de in both walks from per-row constants instead of writing them to [E, H, 2] and
 * gathering them back through posT (cross-lane dot products: results within the 1e-5 / 1e-4 relative
 * parity bar of the GAT op, not bit-identical to the kernels above).
 *   ggl_gat_fast_fwd : as ggl_gat_fused_fwd; N_src = rows of x (selects 32-bit offsets when the panel is < 4 GiB)
 *   ggl_gat_fast_bwd : destination walk (stats[N,H,4] = {er, m, 1/(den + 1e-16), <g_i, out_i>}, ger) then
 *                      source walk (gx, gel).  stats: workspace of N * H * 4 floats; plan->partial >=
 *                      n_chunks * H * 4 DOUBLES, 8-byte aligned (ABI 8: the destination walk keeps four double sums per
 *                      row and head, see gat_fast.hip; = ggl_partial_bytes(GGL_F32, n_chunks, 8 * H, 0));
 *                      planT->partial = ggl_partial_bytes(GGL_F32, n_chunksT, H*C + H, 0);
 *                      posT is only read with p_drop > 0 (the keep bit lives at the forward position). */
int ggl_gat_fast_supported(int64_t H, int64_t C);
int ggl_gat_fast_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el, const float *er,
                     const float *x, int64_t N_src, float slope, int64_t H, int64_t C, float p_drop,
                     int64_t *rng_state, float *out, float *rowmax, float *rowden, void *stream);
int ggl_gat_fast_bwd(const ggl_segplan_t *plan, const int32_t *col, const ggl_segplan_t *planT,
                     const int32_t *colT, const int32_t *posT, const float *el, const float *er,
                     const float *x, const float *g, const float *out, const float *rowmax,
                     const float *rowden, float slope, int64_t H, int64_t C, float p_drop,
                     const int64_t *rng_used, float *stats, float *gx, float *gel, float *ger, void *stream);
/* Head-mean GAT layer aggregated BEFORE it is transformed (the output layer of models/gat.py: concat=False,
 * gat_conv.py:114-122): y_i = 1/H sum_h (sum_j alpha_ijh x_j) W_h, so the per-edge gather is the F-float INPUT row
 * shared by all heads instead of the H x C transformed row (Reddit GAT: 256 B instead of 1408 B), and in the
 * backward the source walk gathers the C-float output gradient g_i (dL/dA_ih = g_i W_h^T / H).  H = 8, F and the
 * padded class width Cp <= 64 (multiples of 4): ggl_gat_sh_supported.  GPU build only (DPP row broadcasts and a
 * 16-lane reduce-scatter); parity bar as for the fused GAT op (1e-5 / 1e-4 relative).
 *   ggl_gat_sh_fwd : rowmax[N,8], A[N,8,F] = sum_j alpha_ijh x_j, den[N,8]; plan->partial =
 *                    ggl_gat_sh_partial_bytes(n_chunks, F) when the plan has long rows
 *   ggl_gat_sh_bwd : ger[N,8] (destination walk: G[N,8,F] = dL/dA and stats[N,8,4] = {er, m, 1/(den+1e-16),
 *                    <G_ih, A_ih>} per row, x gathered) and T[N_src,8,Cp] = sum_i alpha_ijh gy_i, gel[N_src,8]
 *                    (source walk: z[N_src,8,Cp] = the rows' own x_j W_h, gy[N,Cp] gathered); partial buffers:
 *                    plan->partial >= n_chunks * 8 * 4 DOUBLES (ABI 9: the destination walk keeps four row sums in double; ggl_gat_sh_partial_bytes(
 *                    n_chunks, 8) is enough), planT->partial = ggl_gat_sh_partial_bytes(n_chunksT, Cp) */
int ggl_gat_sh_supported(int64_t H, int64_t F, int64_t C);
size_t ggl_gat_sh_partial_bytes(int64_t n_chunks, int64_t F);
int ggl_gat_sh_fwd(const ggl_segplan_t *plan, const int32_t *col, const float *el, const float *er, const float *x,
                   int64_t F, float slope, float p_drop, int64_t *rng_state, float *rowmax, float *A, float *den,
                   void *stream);
int ggl_gat_sh_bwd(const ggl_segplan_t *plan, const int32_t *col, const ggl_segplan_t *planT, const int32_t *colT,
                   const int32_t *posT, const float *el, const float *x, int64_t F, const float *G,
                   const float *stats, const float *z, const float *gy, int64_t Cp, float slope, float p_drop,
                   const int64_t *rng_used, float *ger, float *T, float *gel, void *stream);
/* (ABI 8) the stats panel ggl_gat_sh_bwd reads, in one pass: stats[i,h,:] = {er, rowmax, 1 / (den + 1e-16), <G[i,h,:], A[i,h,:]>};
 * er / rowmax / den [N,8], G / A [N,8,F] (F % 4 == 0, 16-byte aligned), stats [N,8,4] */
int ggl_gat_sh_stats(const float *er, const float *rowmax, const float *den, const float *G, const float *A, int64_t N,
                     int64_t F, float *stats, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Uniform neighbour sampling — supersedes ops/sparse sample_adj
 * (cpu/sample.cpp:10-135).  CSR (rowptr [M+1], col [nnz]) int64; seeds [B] int64 row ids.
 *   ggl_sample_count: out_deg[i] = deg (fanout < 0) | fanout if deg > 0 (replace) | min(deg, fanout)
 *   ggl_sample_pick : given out_rowptr = exclusive prefix of out_deg, writes for every seed its sampled
 *                     positions e_pos (indices into col) and neighbours nbr = col[e_pos]; distinct
 *                     positions by Floyd's algorithm when !replace (sample.cpp:75-83); Philox4x32-10 on
 *                     rng_state = {seed, offset} (device int64[2], offset advanced after the launch).
 * ---------------------------------------------------------------------------------------------- */
int ggl_sample_count(const int64_t *rowptr, const int64_t *seeds, int64_t B, int64_t num_nodes, int64_t fanout, int replace,
                     int64_t *out_deg, void *stream);
int ggl_sample_pick(const int64_t *rowptr, const int64_t *col, const int64_t *seeds, int64_t B,
                    int64_t fanout, int replace, const int64_t *out_rowptr, int64_t *rng_state,
                    int64_t *e_pos, int64_t *nbr, void *stream);
/* One hop with DEVICE-side sizes and fixed capacities — nothing is read back, so a whole mini-batch step
 * (sampling, feature gather, layers, loss, backward, optimizer) captures into one hipGraph.
 *   seeds[B_cap] of which the first *n_seeds_dev are valid; fanout > 0 (min(deg, fanout) distinct neighbours per
 *   seed, Floyd's algorithm as sample.cpp:75-83); first_pos: int64 scratch of one entry per graph node, every
 *   entry == 2^62 on entry and again on exit.
 *   Capacities: E_cap <= B_cap * fanout edge slots, S_cap in [B_cap, B_cap + E_cap] node slots (the worst
 *   case is rarely met: a caller sizes them from measured batches and checks the overflow flag).
 *   out_rowptr[B_cap + 1] (rows past the valid seeds are empty), out_col[E_cap] int32 LOCAL ids, each row
 *   ascending (sample.cpp:112-118), out_eid[E_cap] CSR positions of the sampled edges (or NULL),
 *   out_nid[S_cap]: the seeds verbatim, then the new nodes in first-seen order (sample.cpp:24-55), zero-padded;
 *   out_counts[3] = {nodes in out_nid, sampled edges, overflow} (device): overflow = 1 when a capacity was
 *   hit (rows cut at E_cap / nodes past S_cap dropped — memory-safe, but the block is incomplete). */
size_t ggl_sample_hop_workspace_bytes(int64_t B_cap, int64_t E_cap);
int ggl_sample_hop(const int64_t *rowptr, const int64_t *col, const int64_t *seeds, const int64_t *n_seeds_dev,
                   int64_t B_cap, int64_t num_nodes, int64_t fanout, int64_t E_cap, int64_t S_cap, int64_t *rng_state,
                   int64_t *first_pos, int64_t *out_rowptr, int32_t *out_col, int64_t *out_eid, int64_t *out_nid,
                   int64_t *out_counts, void *workspace, size_t workspace_bytes, void *stream);
/* (ABI 8) the same hop; overflow_total (device, or NULL): += 1 when this hop hit a capacity — the caller's running count
 * of truncated hops, kept by the hop's own kernels instead of a host-side add per hop */
int ggl_sample_hop_ex(const int64_t *rowptr, const int64_t *col, const int64_t *seeds, const int64_t *n_seeds_dev,
                      int64_t B_cap, int64_t num_nodes, int64_t fanout, int64_t E_cap, int64_t S_cap, int64_t *rng_state,
                      int64_t *first_pos, int64_t *out_rowptr, int32_t *out_col, int64_t *out_eid, int64_t *out_nid,
                      int64_t *out_counts, void *workspace, size_t workspace_bytes, int64_t *overflow_total, void *stream);
/* CSC of such a block without a host read: rowptrT[N_src_cap + 1], dstT[E_cap] = destination rows of each source
 * row's block edges (ascending); E_cap entries of col of which rowptr[N_dst] are valid. */
size_t ggl_block_transpose_workspace_bytes(int64_t E_cap, int64_t N_src_cap);
int ggl_block_transpose(const int64_t *rowptr, const int32_t *col, int64_t N_dst, int64_t N_src_cap, int64_t E_cap,
                        int64_t *rowptrT, int32_t *dstT, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Tuning knobs (process-wide; also read once from the environment: GGL_UNROLL, GGL_XCD_SWIZZLE,
 * GGL_FORCE_GENERIC, GGL_ROW_ORDER).  For A/B measurements only — results do not depend on them.
 * ---------------------------------------------------------------------------------------------- */
int ggl_set_option(const char *name, int64_t value);
int64_t ggl_get_option(const char *name);

/* ------------------------------------------------------------------------------------------------
 * Host policy, ONE copy (round 4).  Two hosts drive these kernels — gammagl_amd/ops.py (ctypes) and
 * gammagl_amd/csrc/torch/ggl_torch.cpp (TORCH_LIBRARY) — and both must take the same launch decisions (they are checked
 * against each other bit for bit).  The decisions live here, measured constants and their reasons included; the hosts
 * only ask.  Pure host functions, no device access.
 * ---------------------------------------------------------------------------------------------- */
/* long-row threshold (= elements per chunk) of a plan of E elements: the largest power of two <= E / (256 CUs x 32
 * resident wavefronts), clamped to [256, 4096] — one wavefront walks a row serially, so on a small graph a 4096-element
 * walk is the whole launch (arxiv-sized K = 256 SpMM: 0.685 ms at 4096, 0.284 ms at 256).  GGL_LONG_ROW overrides. */
int64_t ggl_policy_chunk(int64_t E);
/* width a 2-D f32 matrix [N_in, K] is zero-padded to before an SpMM over E edges (== K: as it is).  reduce: 0 = sum /
 * mean, 1 = max.  sum / mean: K > 256 not a multiple of 64 -> next multiple of 64 (whole cache lines per 64-column
 * block: products-sized K = 602 55 -> 42 ms); K % 4 != 0, K >= 8 -> next multiple of 4 (16-byte rows: K = 47 9.3 -> 5.8 ms);
 * max: K > 128 with K % 4 != 0 -> next multiple of 4 (K = 602 69 -> 60 ms).  Only where the copy is paid back:
 * E >= 8 N_in. */
int64_t ggl_policy_spmm_width(int reduce, int64_t K, int64_t E, int64_t N_in);
/* channels per head a [N_in, H, C] tensor is zero-padded to for bspmm / the fused GAT (41 classes -> 44) */
int64_t ggl_policy_head_channels(int64_t C, int64_t E, int64_t N_in);
/* 1: the spmm-mean backward divides the rows of g by their count ONCE and runs the plain transposed SpMM-sum (same
 * bits; products-sized K = 256 23.0 -> 15.5 ms) — where a row has edges to amortise the pass over g: E >= 4 N_in */
int ggl_policy_mean_bwd_prescale(int64_t E, int64_t N_in);
/* form of the gspmm(max) backward over E edges into [N_dst, K] witnesses: 2 = 1-bit winner mask (ggl_spmm_max_mask +
 * ggl_spmm_max_bwd_mask; a transient of ggl_spmm_max_mask_bytes(E, K)), 1 = int32 witness copy (ggl_spmm_max_bwd32),
 * 0 = int64 witnesses (ggl_spmm_max_bwd).  The mask for option maxbwd_mask (128) <= K <= option maxbwd_mask_kmax (256):
 * measured faster on the products- AND the Reddit-sized graph up to K = 256, slower and 96 B per edge at K = 602
 * (profiles/r6_maxbwd_forms.txt).  Hosts also fall back to 1 when the mask cannot be allocated. */
int ggl_policy_maxbwd_form(int64_t E, int64_t N_dst, int64_t K);
/* 1: the bspmm weight gradient walks the destination-sorted plan with LDS-staged strips (edgedot.hip); 0: one thread
 * per (edge, head) in COO order — channel counts that are not multiples of 4, and heads of <= 16 channels unless they
 * are 12 / 16 wide in a row of >= 256 columns (measured on the products-sized graph, forward + backward: 16 x 16
 * 63.7 vs 67.5 ms; 32 x 8 the other way, 82.4 vs 67.0; 8 x 8 equal).  Takes (H, C) since ABI 7. */
int ggl_policy_gradw_sorted(int64_t H, int64_t C);
/* rows per XCD run of a plan (0: round-robin blocks) given the share of its edges whose endpoints lie within N / 64 ids
 * of each other: 2048 for E >= 2^22 and locality > 0.5 (planted-community graph in cluster order: K = 256 13.4 ->
 * 10.8 ms; R-MAT orders lose 7-8 % with it).  GGL_XCD_RUN_ROWS overrides. */
int64_t ggl_policy_xcd_run_rows(int64_t E, double locality);
/* the row hand-out order: rows of >= *heavy elements first, longest first; the rest by length inside windows of
 * *window consecutive ids (0: one global sort).  GGL_ROW_ORDER_WINDOW overrides the window. */
int ggl_policy_row_order(int64_t *window_host, int64_t *heavy_host);

/* Measurement aid (bench.py's roofline leg): a grid-stride pass over `n_vec4` 16-byte vectors of `src`, 16 bytes per
 * lane and four independent loads in flight — mode 0: read-only (every wavefront folds what it read into ONE float of
 * `dst`, which must hold at least 65536 floats), mode 1: copy to `dst`.  The rate it reaches is "what a streaming kernel
 * achieves on this part" next to the 8 TB/s spec peak; no reference counterpart. */
int ggl_calib_stream(const float *src, float *dst, int64_t n_vec4, int mode, void *stream);

/* Profiling aid: run `reps` launches of the dominant SpMM-sum kernel bracketed by hipEvents on
 * `stream` and return the average milliseconds per launch in *ms_host (SYNCHRONOUS). */
int ggl_time_spmm_sum(const ggl_segplan_t *plan, const int32_t *col, const float *w, int w_by_pos,
                      const float *x, int64_t K, float *out, void *stream, int reps,
                      float *ms_host);

#ifdef __cplusplus
}
#endif
#endif /* GGL_MPOPS_H */
