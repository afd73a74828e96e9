/*
 * oracle/ggl_oracle.c — plain-C, single-threaded restatement of the reference CPU mpops.
 * TEST INFRASTRUCTURE ONLY (see ggl_oracle.h).  Compiled with -ffp-contract=off so that
 * `out += w * x` is a rounded multiply followed by a rounded add, exactly like the reference's
 * x86-64 build (no FMA in the baseline ISA g++ targets).
 *
 * Reference paths below are relative to gammagl/mpops/torch_ext/.
 */
#include "ggl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* c10::Half / c10::BFloat16 conversions (round-to-nearest-even; NaN stays NaN)                */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

uint16_t ggl_oracle_f32_to_f16(float f) {
  uint32_t x = f32_bits(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          /* NaN */
  if (ax >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);         /* >= 65536 -> inf */
  if (ax < 0x33000001u) return (uint16_t)sign;                      /* < 2^-25 (or == ) -> 0 */
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  uint32_t half;
  if (e < -14) {                                                    /* subnormal half */
    int shift = -14 - e + 13;                                       /* 14..24 */
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    half = q;
  } else {
    uint32_t q = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;        /* may carry into exponent/inf */
    half = q;
  }
  return (uint16_t)(sign | half);
}

float ggl_oracle_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return bits_f32(sign);
    float v = (float)m * (1.0f / 16777216.0f);                      /* m * 2^-24 */
    return sign ? -v : v;
  }
  if (e == 31) return bits_f32(sign | 0x7f800000u | (m << 13));
  return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

uint16_t ggl_oracle_f32_to_bf16(float f) {
  uint32_t x = f32_bits(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;              /* c10: NaN -> 0x7FC0 */
  uint32_t rounding_bias = ((x >> 16) & 1u) + 0x7fffu;
  return (uint16_t)((x + rounding_bias) >> 16);
}

float ggl_oracle_bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

/* ------------------------------------------------------------------------------------------ */
/* per-dtype scalar semantics: "accumulate in the storage dtype" (segment_sum_cpu.cpp:56)      */
/* ------------------------------------------------------------------------------------------ */
#define INT_OPS(S, T, LOW)                                                              \
  static inline T add_##S(T a, T b) { return (T)(a + b); }                              \
  static inline int less_##S(T a, T b) { return a < b; }                                \
  static inline T lowest_##S(void) { return (T)(LOW); }                                 \
  static inline T one_##S(void) { return (T)1; }                                        \
  static inline T zero_##S(void) { return (T)0; }                                       \
  static inline int gt1_##S(T a) { return a > 1; }                                      \
  static inline T div_##S(T a, T b) { return (T)(a / b); }
INT_OPS(u8, uint8_t, 0)
INT_OPS(i8, int8_t, INT8_MIN)
INT_OPS(i16, int16_t, INT16_MIN)
INT_OPS(i32, int32_t, INT32_MIN)
static inline int64_t add_i64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int less_i64(int64_t a, int64_t b) { return a < b; }
static inline int64_t lowest_i64(void) { return INT64_MIN; }
static inline int64_t one_i64(void) { return 1; }
static inline int64_t zero_i64(void) { return 0; }
static inline int gt1_i64(int64_t a) { return a > 1; }
static inline int64_t div_i64(int64_t a, int64_t b) { return a / b; }

#define FLT_OPS(S, T, LOW)                                                              \
  static inline T add_##S(T a, T b) { return a + b; }                                   \
  static inline int less_##S(T a, T b) { return a < b; }                                \
  static inline T lowest_##S(void) { return LOW; }                                      \
  static inline T one_##S(void) { return (T)1; }                                        \
  static inline T zero_##S(void) { return (T)0; }                                       \
  static inline int gt1_##S(T a) { return a > (T)1; }                                   \
  static inline T div_##S(T a, T b) { return a / b; }
FLT_OPS(f32, float, -FLT_MAX)
FLT_OPS(f64, double, -DBL_MAX)

/* 16-bit floats: every binary op is float(a) op float(b) rounded back (c10/util/Half-inl.h,
 * BFloat16-inl.h operator+ / operator/ / operator<). */
#define H16_OPS(S, TOF, FROMF, LOWBITS)                                                 \
  static inline uint16_t add_##S(uint16_t a, uint16_t b) { return FROMF(TOF(a) + TOF(b)); } \
  static inline int less_##S(uint16_t a, uint16_t b) { return TOF(a) < TOF(b); }        \
  static inline uint16_t lowest_##S(void) { return (uint16_t)(LOWBITS); }               \
  static inline uint16_t one_##S(void) { return FROMF(1.0f); }                          \
  static inline uint16_t zero_##S(void) { return 0; }                                   \
  static inline int gt1_##S(uint16_t a) { return TOF(a) > 1.0f; }                       \
  static inline uint16_t div_##S(uint16_t a, uint16_t b) { return FROMF(TOF(a) / TOF(b)); }
H16_OPS(f16, ggl_oracle_f16_to_f32, ggl_oracle_f32_to_f16, 0xFBFFu)
H16_OPS(bf16, ggl_oracle_bf16_to_f32, ggl_oracle_f32_to_bf16, 0xFF7Fu)

static int check_index(const int64_t *idx, int64_t E, int64_t N) {
  for (int64_t e = 0; e < E; ++e)
    if (idx[e] < 0 || idx[e] >= N) return GGL_ORACLE_EINDEX;
  return GGL_ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* segment_sum: cpu/segment_sum_cpu.cpp:11-60 (hot loop :47-56)                                */
/* segment_mean: cpu/segment_mean_cpu.cpp:11-80 — sums as above, a per-segment count held in   */
/*   x's dtype (:44,52), then out[s] /= count[s] where count[s] > 1 (:67-76).  The reference    */
/*   sizes the count buffer by E and only divides rows s < E; for ids >= E it writes out of    */
/*   bounds.  On its defined domain (max(ids) < E) the statement below is identical; outside   */
/*   it this oracle gives the mathematically intended mean.             */
/* segment_max: cpu/segment_max_cpu.cpp:11-69 — out pre-filled with lowest() (:40), strict <   */
/*   in serial order (:59-62) so the smallest e wins ties and NaN never wins; if x is empty    */
/*   the function returns zeros before the fill (:28-30).                                      */
/* ------------------------------------------------------------------------------------------ */
#define GEN_SEGMENT(S, T)                                                               \
  static void seg_sum_##S(const T *x, const int64_t *idx, int64_t E, int64_t K,         \
                          int64_t N, T *out) {                                          \
    for (int64_t i = 0; i < N * K; ++i) out[i] = zero_##S();                            \
    for (int64_t e = 0; e < E; ++e) {                                                   \
      int64_t s = idx[e];                                                               \
      for (int64_t k = 0; k < K; ++k) out[s * K + k] = add_##S(out[s * K + k], x[e * K + k]); \
    }                                                                                   \
  }                                                                                     \
  static int seg_mean_##S(const T *x, const int64_t *idx, int64_t E, int64_t K,         \
                          int64_t N, T *out) {                                          \
    T *deg = (T *)malloc(sizeof(T) * (size_t)(N > 0 ? N : 1));                          \
    if (!deg) return GGL_ORACLE_EDTYPE;                                                 \
    for (int64_t s = 0; s < N; ++s) deg[s] = zero_##S();                                \
    for (int64_t i = 0; i < N * K; ++i) out[i] = zero_##S();                            \
    for (int64_t e = 0; e < E; ++e) {                                                   \
      int64_t s = idx[e];                                                               \
      deg[s] = add_##S(deg[s], one_##S());                                              \
      for (int64_t k = 0; k < K; ++k) out[s * K + k] = add_##S(out[s * K + k], x[e * K + k]); \
    }                                                                                   \
    for (int64_t s = 0; s < N; ++s)                                                     \
      if (gt1_##S(deg[s]))                                                              \
        for (int64_t k = 0; k < K; ++k) out[s * K + k] = div_##S(out[s * K + k], deg[s]); \
    free(deg);                                                                          \
    return GGL_ORACLE_OK;                                                               \
  }                                                                                     \
  static void seg_max_##S(const T *x, const int64_t *idx, int64_t E, int64_t K,         \
                          int64_t N, T *out, int64_t *arg, int64_t arg_fill) {          \
    for (int64_t i = 0; i < N * K; ++i) arg[i] = arg_fill;                              \
    if (E * K == 0) {                                                                   \
      for (int64_t i = 0; i < N * K; ++i) out[i] = zero_##S();                          \
      return;                                                                           \
    }                                                                                   \
    for (int64_t i = 0; i < N * K; ++i) out[i] = lowest_##S();                          \
    for (int64_t e = 0; e < E; ++e) {                                                   \
      int64_t s = idx[e];                                                               \
      for (int64_t k = 0; k < K; ++k) {                                                 \
        T cur = x[e * K + k];                                                           \
        if (less_##S(out[s * K + k], cur)) {                                            \
          out[s * K + k] = cur;                                                         \
          arg[s * K + k] = e;                                                           \
        }                                                                               \
      }                                                                                 \
    }                                                                                   \
  }
GEN_SEGMENT(u8, uint8_t)
GEN_SEGMENT(i8, int8_t)
GEN_SEGMENT(i16, int16_t)
GEN_SEGMENT(i32, int32_t)
GEN_SEGMENT(i64, int64_t)
GEN_SEGMENT(f16, uint16_t)
GEN_SEGMENT(bf16, uint16_t)
GEN_SEGMENT(f32, float)
GEN_SEGMENT(f64, double)

#define DISPATCH(dtype, CALL)                                                           \
  switch (dtype) {                                                                      \
    case GGL_U8: { typedef uint8_t T; CALL(u8); break; }                                \
    case GGL_I8: { typedef int8_t T; CALL(i8); break; }                                 \
    case GGL_I16: { typedef int16_t T; CALL(i16); break; }                              \
    case GGL_I32: { typedef int32_t T; CALL(i32); break; }                              \
    case GGL_I64: { typedef int64_t T; CALL(i64); break; }                              \
    case GGL_F16: { typedef uint16_t T; CALL(f16); break; }                             \
    case GGL_BF16: { typedef uint16_t T; CALL(bf16); break; }                           \
    case GGL_F32: { typedef float T; CALL(f32); break; }                                \
    case GGL_F64: { typedef double T; CALL(f64); break; }                               \
    default: return GGL_ORACLE_EDTYPE;                                                  \
  }

int ggl_oracle_segment_sum(int dtype, const void *x, const int64_t *idx, int64_t E, int64_t K,
                           int64_t N, void *out) {
  if (check_index(idx, E, N)) return GGL_ORACLE_EINDEX;
#define CALL(S) seg_sum_##S((const T *)x, idx, E, K, N, (T *)out)
  DISPATCH(dtype, CALL)
#undef CALL
  return GGL_ORACLE_OK;
}

int ggl_oracle_segment_mean(int dtype, const void *x, const int64_t *idx, int64_t E, int64_t K,
                            int64_t N, void *out) {
  if (check_index(idx, E, N)) return GGL_ORACLE_EINDEX;
  int rc = GGL_ORACLE_OK;
#define CALL(S) rc = seg_mean_##S((const T *)x, idx, E, K, N, (T *)out)
  DISPATCH(dtype, CALL)
#undef CALL
  return rc;
}

int ggl_oracle_segment_max(int dtype, const void *x, const int64_t *idx, int64_t E, int64_t K,
                           int64_t N, void *out, int64_t *arg, int64_t arg_fill) {
  /* segment_max_cpu.cpp:50 raises IndexError for idx >= N */
  if (check_index(idx, E, N)) return GGL_ORACLE_EINDEX;
#define CALL(S) seg_max_##S((const T *)x, idx, E, K, N, (T *)out, arg, arg_fill)
  DISPATCH(dtype, CALL)
#undef CALL
  return GGL_ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* backward passes of the segment ops (autograd glue in src/)                                  */
/* ------------------------------------------------------------------------------------------ */
/* src/segment_sum.cpp:43-54: grad_in = grad_out.index_select(0, index) */
int ggl_oracle_segment_sum_bwd(int dtype, const void *gout, const int64_t *idx, int64_t E,
                               int64_t K, int64_t N, void *gin) {
  if (check_index(idx, E, N)) return GGL_ORACLE_EINDEX;
  size_t es = dtype == GGL_F32 ? 4 : dtype == GGL_F64 ? 8 : 0;
  if (!es) return GGL_ORACLE_EDTYPE;
  for (int64_t e = 0; e < E; ++e)
    memcpy((char *)gin + (size_t)e * K * es, (const char *)gout + (size_t)idx[e] * K * es,
           (size_t)K * es);
  return GGL_ORACLE_OK;
}

/* src/segment_mean.cpp:44-63: grad_out[index] / bincount(index)[index] (int64 count -> float) */
int ggl_oracle_segment_mean_bwd(int dtype, const void *gout, const int64_t *idx, int64_t E,
                                int64_t K, int64_t N, void *gin) {
  if (check_index(idx, E, N)) return GGL_ORACLE_EINDEX;
  if (dtype != GGL_F32 && dtype != GGL_F64) return GGL_ORACLE_EDTYPE;
  int64_t *cnt = (int64_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) cnt[idx[e]]++;
  for (int64_t e = 0; e < E; ++e) {
    int64_t s = idx[e];
    for (int64_t k = 0; k < K; ++k) {
      if (dtype == GGL_F32)
        ((float *)gin)[e * K + k] = ((const float *)gout)[s * K + k] / (float)cnt[s];
      else
        ((double *)gin)[e * K + k] = ((const double *)gout)[s * K + k] / (double)cnt[s];
    }
  }
  free(cnt);
  return GGL_ORACLE_OK;
}

/* src/segment_max.cpp:48-61: zeros[E+1,K].scatter_(0, arg, grad_out)[:E].  The reference's
 * "empty" sentinel is N, which aliases a real row when N < E and overflows when N > E
 * (SURVEY.md §8a row A3); here any arg >= E means "no gradient". */
int ggl_oracle_segment_max_bwd(int dtype, const void *gout, const int64_t *arg, int64_t E,
                               int64_t K, int64_t N, void *gin) {
  size_t es = dtype == GGL_F32 ? 4 : dtype == GGL_F64 ? 8 : 0;
  if (!es) return GGL_ORACLE_EDTYPE;
  memset(gin, 0, (size_t)E * K * es);
  for (int64_t s = 0; s < N; ++s)
    for (int64_t k = 0; k < K; ++k) {
      int64_t a = arg[s * K + k];
      if (a < 0 || a >= E) continue;
      memcpy((char *)gin + ((size_t)a * K + k) * es, (const char *)gout + ((size_t)s * K + k) * es,
             es);
    }
  return GGL_ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* gspmm: COO SpMM, f32 only, out = zeros_like(x)                                              */
/* ------------------------------------------------------------------------------------------ */
static int check_coo(const int64_t *index, int64_t E, int64_t N) {
  for (int64_t e = 0; e < 2 * E; ++e)
    if (index[e] < 0 || index[e] >= N) return GGL_ORACLE_EINDEX;
  return GGL_ORACLE_OK;
}

/* cpu/spmm_sum_cpu.cpp:5-41 (loop :29-39): out[dst] += w[e] * x[src] */
int ggl_oracle_spmm_sum_fwd(const int64_t *index, const float *w, const float *x, int64_t E,
                            int64_t N, int64_t K, float *out) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * K; ++i) out[i] = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t k = 0; k < K; ++k) out[dst * K + k] += w[e] * x[src * K + k];
  }
  return GGL_ORACLE_OK;
}

/* cpu/spmm_sum_cpu.cpp:43-80: gx[src] += w[e] * g[dst]  (no gradient for w: gspmm.cpp:30) */
int ggl_oracle_spmm_sum_bwd(const int64_t *index, const float *w, const float *g, int64_t E,
                            int64_t N, int64_t K, float *gx) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * K; ++i) gx[i] = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t k = 0; k < K; ++k) gx[src * K + k] += w[e] * g[dst * K + k];
  }
  return GGL_ORACLE_OK;
}

/* cpu/spmm_mean_cpu.cpp:5-61: weighted sum divided by the UNWEIGHTED in-degree (:27,37,51-58) */
int ggl_oracle_spmm_mean_fwd(const int64_t *index, const float *w, const float *x, int64_t E,
                             int64_t N, int64_t K, float *out, int64_t *count) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * K; ++i) out[i] = 0.0f;
  for (int64_t n = 0; n < N; ++n) count[n] = 0;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    count[dst]++;
    for (int64_t k = 0; k < K; ++k) out[dst * K + k] += w[e] * x[src * K + k];
  }
  for (int64_t n = 0; n < N; ++n)
    if (count[n] > 0)
      for (int64_t k = 0; k < K; ++k) out[n * K + k] /= (float)count[n];
  return GGL_ORACLE_OK;
}

/* cpu/spmm_mean_cpu.cpp:63-105 (:95-101): gx[src] += g[dst] / count[dst] * w[e] */
int ggl_oracle_spmm_mean_bwd(const int64_t *index, const float *w, const float *g,
                             const int64_t *count, int64_t E, int64_t N, int64_t K, float *gx) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * K; ++i) gx[i] = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t k = 0; k < K; ++k) {
      float c = g[dst * K + k] / (float)count[dst] * w[e];
      gx[src * K + k] += c;
    }
  }
  return GGL_ORACLE_OK;
}

/* cpu/spmm_max_cpu.cpp:5-55: out pre-filled with lowest() (:18-19), argmax stored as the SRC node
 * id (:47), strict < in serial order */
int ggl_oracle_spmm_max_fwd(const int64_t *index, const float *w, const float *x, int64_t E,
                            int64_t N, int64_t K, float *out, int64_t *argsrc) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * K; ++i) { out[i] = -FLT_MAX; argsrc[i] = 0; }
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t k = 0; k < K; ++k) {
      float v = w[e] * x[src * K + k];
      if (out[dst * K + k] < v) { out[dst * K + k] = v; argsrc[dst * K + k] = src; }
    }
  }
  return GGL_ORACLE_OK;
}

/* cpu/spmm_max_cpu.cpp:57-99 (:88-93): every edge whose src equals the stored id gets gradient */
int ggl_oracle_spmm_max_bwd(const int64_t *index, const float *w, const float *g,
                            const int64_t *argsrc, int64_t E, int64_t N, int64_t K, float *gx) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * K; ++i) gx[i] = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t k = 0; k < K; ++k)
      if (argsrc[dst * K + k] == src) gx[src * K + k] += w[e] * g[dst * K + k];
  }
  return GGL_ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* bspmm: cpu/bspmm_sum_cpu.cpp:7-56 (fwd :40-54), :58-113 (bwd :88-108)                       */
/* ------------------------------------------------------------------------------------------ */
int ggl_oracle_bspmm_sum_fwd(const int64_t *index, const float *w, const float *x, int64_t E,
                             int64_t N, int64_t H, int64_t C, float *out) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * H * C; ++i) out[i] = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t h = 0; h < H; ++h)
      for (int64_t c = 0; c < C; ++c)
        out[dst * C * H + h * C + c] += w[e * H + h] * x[src * C * H + h * C + c];
  }
  return GGL_ORACLE_OK;
}

int ggl_oracle_bspmm_sum_bwd(const int64_t *index, const float *w, const float *x, const float *g,
                             int64_t E, int64_t N, int64_t H, int64_t C, float *gx, float *gw) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  for (int64_t i = 0; i < N * H * C; ++i) gx[i] = 0.0f;
  for (int64_t i = 0; i < E * H; ++i) gw[i] = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    int64_t src = index[e], dst = index[e + E];
    for (int64_t h = 0; h < H; ++h)
      for (int64_t c = 0; c < C; ++c) {
        gx[src * C * H + h * C + c] += w[e * H + h] * g[dst * C * H + h * C + c];
        gw[e * H + h] += x[src * C * H + h * C + c] * g[dst * C * H + h * C + c];
      }
  }
  return GGL_ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* GAT edge-softmax + aggregate (gat_conv.py:103-112, utils/softmax.py:29-35), composed from    */
/* the oracle's own segment ops so every quirk (lowest() fill, serial sums) carries over.       */
/* ------------------------------------------------------------------------------------------ */
int ggl_oracle_gat_fwd(const int64_t *index, const float *el, const float *er, const float *x,
                       float slope, int64_t E, int64_t N, int64_t H, int64_t C, float *out,
                       float *alpha_out) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  const int64_t *src = index, *dst = index + E;
  size_t eh = (size_t)(E * H > 0 ? E * H : 1), nh = (size_t)(N * H > 0 ? N * H : 1);
  float *s = (float *)calloc(eh, 4), *ex = (float *)malloc(eh * 4);
  float *m = (float *)malloc(nh * 4), *d = (float *)malloc(nh * 4);
  int64_t *arg = (int64_t *)malloc(nh * 8);
  float *msg = (float *)malloc((size_t)(E * H * C > 0 ? E * H * C : 1) * 4);
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) {
      float v = el[src[e] * H + h] + er[dst[e] * H + h];
      s[e * H + h] = v > 0.0f ? v : v * slope;                     /* tlx LeakyReLU == torch */
    }
  seg_max_f32(s, dst, E, H, N, m, arg, E);
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) ex[e * H + h] = expf(s[e * H + h] - m[dst[e] * H + h]);
  seg_sum_f32(ex, dst, E, H, N, d);
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) {
      float a = ex[e * H + h] / (d[dst[e] * H + h] + 1e-16f);
      if (alpha_out) alpha_out[e * H + h] = a;
      for (int64_t c = 0; c < C; ++c)
        msg[(e * H + h) * C + c] = x[(src[e] * H + h) * C + c] * a;  /* message_passing.py:56-59 */
    }
  seg_sum_f32(msg, dst, E, H * C, N, out);
  free(s); free(ex); free(m); free(d); free(arg); free(msg);
  return GGL_ORACLE_OK;
}

int ggl_oracle_gat_bwd(const int64_t *index, const float *el, const float *er, const float *x,
                       const float *g, float slope, int64_t E, int64_t N, int64_t H, int64_t C,
                       float *gel, float *ger, float *gx) {
  if (check_coo(index, E, N)) return GGL_ORACLE_EINDEX;
  const int64_t *src = index, *dst = index + E;
  size_t nh = (size_t)(N * H > 0 ? N * H : 1), eh = (size_t)(E * H > 0 ? E * H : 1);
  double *m = (double *)malloc(nh * 8), *d = (double *)calloc(nh, 8), *dot = (double *)calloc(nh, 8);
  double *al = (double *)malloc(eh * 8), *da = (double *)malloc(eh * 8);
  double *Gel = (double *)calloc(nh, 8), *Ger = (double *)calloc(nh, 8);
  double *Gx = (double *)calloc((size_t)(N * H * C > 0 ? N * H * C : 1), 8);
  for (size_t i = 0; i < nh; ++i) m[i] = -DBL_MAX;
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) {
      double v = (double)el[src[e] * H + h] + (double)er[dst[e] * H + h];
      double sv = v > 0 ? v : v * (double)slope;
      al[e * H + h] = sv;
      if (m[dst[e] * H + h] < sv) m[dst[e] * H + h] = sv;
    }
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) {
      al[e * H + h] = exp(al[e * H + h] - m[dst[e] * H + h]);
      d[dst[e] * H + h] += al[e * H + h];
    }
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) {
      al[e * H + h] /= (d[dst[e] * H + h] + 1e-16);
      double acc = 0;
      for (int64_t c = 0; c < C; ++c)
        acc += (double)g[(dst[e] * H + h) * C + c] * (double)x[(src[e] * H + h) * C + c];
      da[e * H + h] = acc;
      dot[dst[e] * H + h] += al[e * H + h] * acc;
    }
  for (int64_t e = 0; e < E; ++e)
    for (int64_t h = 0; h < H; ++h) {
      double a = al[e * H + h];
      double ds = a * (da[e * H + h] - dot[dst[e] * H + h]);
      double v = (double)el[src[e] * H + h] + (double)er[dst[e] * H + h];
      double dv = ds * (v > 0 ? 1.0 : (double)slope);
      Gel[src[e] * H + h] += dv;
      Ger[dst[e] * H + h] += dv;
      for (int64_t c = 0; c < C; ++c)
        Gx[(src[e] * H + h) * C + c] += a * (double)g[(dst[e] * H + h) * C + c];
    }
  for (int64_t i = 0; i < N * H; ++i) { gel[i] = (float)Gel[i]; ger[i] = (float)Ger[i]; }
  for (int64_t i = 0; i < N * H * C; ++i) gx[i] = (float)Gx[i];
  free(m); free(d); free(dot); free(al); free(da); free(Gel); free(Ger); free(Gx);
  return GGL_ORACLE_OK;
}
