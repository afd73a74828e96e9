// gammagl_amd/csrc/host/gpu_only_stubs.cpp — the host build (host_shim.hpp) cannot run the kernels of
// gammagl_amd/csrc/gat_fast.hip (they exchange values between lanes); it exports the same symbols so that the ctypes
// binding loads, answers "not supported" to the capability queries and fails loudly if a path is called anyway.
#include <stddef.h>
#include <stdint.h>

#include "../../../include/ggl_mpops.h"

namespace ggl { void set_error(const char *fmt, ...); }

extern "C" int ggl_gat_fast_supported(int64_t, int64_t) { return 0; }
extern "C" int ggl_gat_sh_supported(int64_t, int64_t, int64_t) { return 0; }
extern "C" size_t ggl_gat_sh_partial_bytes(int64_t n_chunks, int64_t F) {
  return n_chunks <= 0 ? 0 : (size_t)n_chunks * (size_t)(8 * F + 16) * sizeof(float) + 64;
}
static int no_gpu() {
  ggl::set_error("this GAT path exists in the GPU build only");
  return GGL_EINVAL;
}
extern "C" int ggl_gat_fast_fwd(const ggl_segplan_t *, const int32_t *, const float *, const float *, const float *,
                                int64_t, float, int64_t, int64_t, float, int64_t *, float *, float *, float *, void *) {
  return no_gpu();
}
extern "C" int ggl_gat_fast_bwd(const ggl_segplan_t *, const int32_t *, const ggl_segplan_t *, const int32_t *,
                                const int32_t *, const float *, const float *, const float *, const float *,
                                const float *, const float *, const float *, float, int64_t, int64_t, float,
                                const int64_t *, float *, float *, float *, float *, void *) {
  return no_gpu();
}
extern "C" int ggl_gat_sh_fwd(const ggl_segplan_t *, const int32_t *, const float *, const float *, const float *,
                              int64_t, float, float, int64_t *, float *, float *, float *, void *) {
  return no_gpu();
}
extern "C" int ggl_gat_sh_stats(const float *, const float *, const float *, const float *, const float *, int64_t, int64_t,
                                float *, void *) {
  return no_gpu();
}
extern "C" int ggl_gat_sh_bwd(const ggl_segplan_t *, const int32_t *, const ggl_segplan_t *, const int32_t *,
                              const int32_t *, const float *, const float *, int64_t, const float *, const float *,
                              const float *, const float *, int64_t, float, float, const int64_t *, float *, float *,
                              float *, void *) {
  return no_gpu();
}

// gammagl_amd/csrc/hub16.hip (LDS + barriers) has no host-emulated build either: the emulated library reports the
// path as unsupported and the 16-bit sums keep walking their hub rows with the row kernel of reduce.hip.
extern "C" int ggl_segment_hub16_supported(int, int64_t, const void *, const void *) { return 0; }
extern "C" int ggl_segment_hub16(int, int, const void *, const ggl_segplan_t *, int64_t, void *, void *) { return no_gpu(); }
