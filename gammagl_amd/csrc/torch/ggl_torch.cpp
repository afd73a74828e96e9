// libggl_torch.so — TORCH_LIBRARY(ggl, ...): the seven operators of gammagl/mpops/torch_ext registered with the
// PyTorch dispatcher FROM C++, the style the reference names as intended (docs/.../register_cpp_ops.md:29-31) but does
// not use (src/operators.cpp:51-59 binds pybind11 free functions).  An op call is dispatcher -> this file -> C ABI
// (include/ggl_mpops.h) -> kernel: no Python frame on the path, callable from C++ / TorchScript / torch.compile.
//
//   ggl::segment_sum / segment_mean (Tensor x, Tensor index, int N) -> Tensor         src/segment_{sum,mean}.cpp
//   ggl::segment_max                (Tensor x, Tensor index, int N) -> (Tensor, Tensor)  src/segment_max.cpp (+ argmax)
//   ggl::spmm_sum / spmm_mean / spmm_max (Tensor index, Tensor? weight, Tensor x) -> Tensor   src/gspmm.cpp:26-202
//   ggl::bspmm_sum                  (Tensor index, Tensor weight, Tensor x) -> Tensor   src/gspmm.cpp:204-260
//
// Backend keys: CUDA (= HIP on ROCm) -> libggl_mpops_hip.so, CPU -> libggl_mpops_host.so (the host build of the same
// kernel sources), both resolved with dlopen from the directory this library sits in — a missing kernel library is a
// loud error at the first call of that device, nothing is computed anywhere else.  Autograd: torch::autograd::Function
// per op, same formulas as the reference's (segment_sum.cpp:43-54, segment_mean.cpp:44-63, segment_max.cpp:48-61,
// gspmm.cpp:57-80,...).  Meta: shapes only.
//
// What lives here besides the registrations is the host side of the op library, the same one gammagl_amd/ops.py
// implements (the two are checked against each other bit for bit, tests/test_torch_cpp.py).  The launch DECISIONS —
// thresholds, padded widths, which walk a gradient takes — are not written twice: both hosts ask the kernel library
// (ggl_policy_*, include/ggl_mpops.h); what each host owns is the plumbing around them:
//   * plan cache keyed on (storage, offset, shape, version, N) of the id tensor, entries die with the storage;
//   * long-row threshold = f(E) (auto_chunk), row hand-out order in id windows, XCD runs for graphs with locality;
//   * rows of K % 4 != 0 / K > 256 && K % 64 != 0 floats aggregate on one padded copy;
//   * edge weights seen twice on a plan are kept in sorted order (w_by_pos);
//   * f16 / bf16 sums walk rows in one piece, hub rows through ggl_segment_hub16 where the build has it;
//   * spmm-mean backward = rows pre-divided by their count + the plain transposed SpMM-sum.
#include <ATen/ATen.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <string>

#include "ggl_mpops.h"

namespace ggl_torch {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---------------------------------------------------------------------------------------------------------------
// The C ABI of one build of the kernel library, resolved at run time (both builds export the same names).
// ---------------------------------------------------------------------------------------------------------------
#define GGL_FNS(X)                                                                                                  \
  X(ggl_abi_version) X(ggl_last_error) X(ggl_plan_workspace_bytes) X(ggl_plan_build) X(ggl_plan_long_workspace_bytes) \
  X(ggl_plan_long_count) X(ggl_plan_long_fill) X(ggl_partial_bytes) X(ggl_gather_i64_to_i32) X(ggl_gather_rows_f32)  \
  X(ggl_segment_sum) X(ggl_segment_mean) X(ggl_segment_max) X(ggl_segment_hub16_supported) X(ggl_segment_hub16)      \
  X(ggl_segment_sum_bwd) X(ggl_segment_mean_bwd) X(ggl_segment_max_bwd) X(ggl_spmm_sum) X(ggl_spmm_mean)             \
  X(ggl_spmm_max) X(ggl_spmm_mean_bwd) X(ggl_spmm_max_bwd) X(ggl_bspmm_sum) X(ggl_bspmm_grad_w)                      \
  X(ggl_bspmm_grad_w_sorted_scratch_bytes) X(ggl_bspmm_grad_w_sorted)                                                \
  X(ggl_gat_partial_bytes) X(ggl_gat_fused_fwd) X(ggl_gat_fused_bwd_dst) X(ggl_gat_fused_bwd_src)                      \
  X(ggl_gat_fast_supported) X(ggl_gat_fast_fwd) X(ggl_gat_fast_bwd) X(ggl_bias_act_fwd)                                \
  X(ggl_bias_act_bwd_workspace_bytes) X(ggl_bias_act_bwd) X(ggl_spmm_epi_ex) X(ggl_segment_epi)                        \
  X(ggl_sample_hop_workspace_bytes) X(ggl_sample_hop)                                                                \
  X(ggl_policy_chunk) X(ggl_policy_spmm_width) X(ggl_policy_head_channels) X(ggl_policy_mean_bwd_prescale)            \
  X(ggl_policy_gradw_sorted) X(ggl_policy_xcd_run_rows) X(ggl_policy_row_order)                                     \
  X(ggl_spmm_max_mask_bytes) X(ggl_spmm_max_mask) X(ggl_spmm_max_bwd_mask) X(ggl_invert_perm) X(ggl_get_option)      \
  X(ggl_spmm_max_bwd32) X(ggl_policy_maxbwd_form)

struct Api {
  void *handle = nullptr;
  std::string path;
#define X(n) decltype(&::n) n = nullptr;
  GGL_FNS(X)
#undef X
};

static std::string here() {
  Dl_info info;
  if (dladdr(reinterpret_cast<void *>(&here), &info) == 0 || info.dli_fname == nullptr) return ".";
  std::string p(info.dli_fname);
  auto slash = p.rfind('/');
  return slash == std::string::npos ? "." : p.substr(0, slash);
}

static Api load(const char *env, const char *file, const char *what) {
  Api a;
  const char *over = std::getenv(env);
  a.path = over != nullptr ? std::string(over) : here() + "/" + file;
  a.handle = dlopen(a.path.c_str(), RTLD_NOW | RTLD_LOCAL);
  TORCH_CHECK(a.handle != nullptr, "gammagl_amd: cannot load ", a.path, " (", dlerror(), "): ", what,
              " tensors are served by this library only — build it with `make -C gammagl_amd/csrc`");
#define X(n)                                                                     \
  a.n = reinterpret_cast<decltype(a.n)>(dlsym(a.handle, #n));                    \
  TORCH_CHECK(a.n != nullptr, "gammagl_amd: ", a.path, " does not export " #n);
  GGL_FNS(X)
#undef X
  TORCH_CHECK(a.ggl_abi_version() == GGL_ABI_VERSION, "gammagl_amd: ", a.path, " has C ABI version ",
              a.ggl_abi_version(), ", this binding was built against ", GGL_ABI_VERSION);
  return a;
}

static const Api &api_for(const c10::Device &d) {
  if (d.is_cuda()) {
    static const Api hip = load("GGL_TORCH_HIP_LIB", "libggl_mpops_hip.so", "GPU");
    return hip;
  }
  TORCH_CHECK(d.is_cpu(), "gammagl_amd: no kernels for device ", d);
  static const Api host = load("GGL_TORCH_HOST_LIB", "libggl_mpops_host.so", "CPU");
  return host;
}

static void check(const Api &a, int rc) {
  if (rc == GGL_OK) return;
  const char *m = a.ggl_last_error();
  std::string msg(m != nullptr ? m : "");
  TORCH_CHECK_INDEX(rc != GGL_EINDEX, msg);
  TORCH_CHECK(false, "ggl_mpops error ", rc, ": ", msg);
}

static void *stream_of(const c10::Device &d) {
  return d.is_cuda() ? static_cast<void *>(c10::hip::getCurrentHIPStream(d.index()).stream()) : nullptr;
}

static int dtype_code(const Tensor &t) {
  switch (t.scalar_type()) {
    case at::kByte: return GGL_U8;
    case at::kChar: return GGL_I8;
    case at::kShort: return GGL_I16;
    case at::kInt: return GGL_I32;
    case at::kLong: return GGL_I64;
    case at::kHalf: return GGL_F16;
    case at::kBFloat16: return GGL_BF16;
    case at::kFloat: return GGL_F32;
    case at::kDouble: return GGL_F64;
    default: TORCH_CHECK(false, "unsupported dtype ", t.scalar_type());
  }
}

static void same_device(std::initializer_list<const Tensor *> ts) {
  const Tensor *first = nullptr;
  for (const Tensor *t : ts) {
    if (t == nullptr || !t->defined()) continue;
    if (first == nullptr) first = t;
    TORCH_CHECK(t->device() == first->device(), "Tensor device inconsistent error.");   // segment_sum.cpp:31
  }
}

static int64_t env_int(const char *name, int64_t dflt) {
  const char *v = std::getenv(name);
  return v != nullptr ? std::strtoll(v, nullptr, 10) : dflt;
}

// ---------------------------------------------------------------------------------------------------------------
// Plans (struct ggl_segplan + the tensors that own its arrays)
// ---------------------------------------------------------------------------------------------------------------
static int64_t auto_chunk(const Api &a, int64_t E) { return a.ggl_policy_chunk(E); }   // (GGL_LONG_ROW is read there)

static Tensor row_order_of(const Tensor &counts);

static bool capturing(const c10::Device &d) {
  if (!d.is_cuda()) return false;
  c10::OptionalDeviceGuard guard(d);
  return c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None;
}

struct SegPlan {
  int64_t N = 0, E = 0, chunk = 0, n_long = 0, n_chunks = 0, max_len = 0, xcd_run = 0;
  bool hub_first = false;   // the long rows lead the id range (a degree-sorted node order): ggl_segplan.xcd_run_rows = -1
  bool sorted = false;
  uint64_t uid = 0;
  Tensor rowptr, perm, long_rows, chunk_ptr, long_order;
  // the row hand-out order is a scheduling aid worth ~100 us of sorting: computed when the plan is launched a SECOND
  // time, so a plan used once (a fresh edge list per mini-batch) never pays for it (ops.py SegPlan.c_struct)
  mutable Tensor row_order;
  mutable std::mutex order_mu;
  mutable int uses = 0;

  // `unsplit`: every row walked in one piece; `skip_long`: rows longer than chunk left to ggl_segment_hub16
  ggl_segplan_t c(const Tensor &partial, bool unsplit = false, bool skip_long = false) const {
    ggl_segplan_t s{};
    const bool lng = n_long > 0 && !unsplit && !skip_long;
    s.rowptr = rowptr.data_ptr<int64_t>();
    s.perm = perm.defined() ? perm.data_ptr<int32_t>() : nullptr;
    s.long_rows = lng ? long_rows.data_ptr<int32_t>() : nullptr;
    s.chunk_ptr = lng ? chunk_ptr.data_ptr<int64_t>() : nullptr;
    s.n_long = lng ? n_long : 0;
    s.n_chunks = lng ? n_chunks : 0;
    s.chunk = unsplit ? (int64_t(1) << 62) : chunk;
    s.partial = partial.defined() ? partial.data_ptr() : nullptr;
    s.N = N;
    s.E = E;
    if (N > 1) {
      std::lock_guard<std::mutex> g(order_mu);
      // (not while a hipGraph is being recorded: the sort would be allocated in the capture pool and filled on replay only)
      if (!row_order.defined() && ++uses >= 2 && !capturing(rowptr.device())) row_order = row_order_of(counts());
    }
    s.row_order = row_order.defined() ? row_order.data_ptr<int32_t>() : nullptr;
    s.xcd_run_rows = xcd_run > 0 ? xcd_run : ((lng && hub_first) ? -1 : 0);
    s.long_order = (lng && long_order.defined()) ? long_order.data_ptr<int32_t>() : nullptr;
    s.max_len = max_len;
    return s;
  }
  Tensor counts() const { return rowptr.slice(0, 1) - rowptr.slice(0, 0, N); }
};

static std::atomic<uint64_t> g_plans_built{0}, g_plan_hits{0};

// rows by descending length inside windows of consecutive ids, the heavy ones first (ops.py Engine._row_order)
static Tensor row_order_of(const Tensor &counts) {
  int64_t W = 2048, heavy = 1024;
  api_for(counts.device()).ggl_policy_row_order(&W, &heavy);
  const int64_t N = counts.size(0);
  if (W <= 0 || N <= W) return at::argsort(counts, /*stable=*/true, 0, /*descending=*/true).to(at::kInt);
  Tensor ar = at::arange(N, counts.options());
  Tensor group = at::where(counts >= heavy, at::zeros_like(ar), at::floor_divide(ar, W) + 1);
  Tensor key = at::bitwise_or(at::bitwise_left_shift(group, 32),
                              (int64_t(1) << 31) - counts.clamp_max((int64_t(1) << 31) - 1));
  return at::argsort(key, /*stable=*/true).to(at::kInt);
}

static void fill_long_rows(const Api &a, SegPlan &p, void *st) {
  if (p.max_len <= p.chunk) return;
  auto bytes = p.rowptr.options().dtype(at::kByte);
  const size_t lwb = a.ggl_plan_long_workspace_bytes(p.N);
  Tensor lws = at::empty({static_cast<int64_t>(lwb)}, bytes);
  int64_t nl = 0, nc = 0;
  check(a, a.ggl_plan_long_count(p.rowptr.data_ptr<int64_t>(), p.N, p.chunk, lws.data_ptr(), lwb, st, &nl, &nc));
  p.n_long = nl;
  p.n_chunks = nc;
  p.long_rows = at::empty({nl}, p.rowptr.options().dtype(at::kInt));
  p.chunk_ptr = at::empty({nl + 1}, p.rowptr.options());
  check(a, a.ggl_plan_long_fill(p.rowptr.data_ptr<int64_t>(), p.N, p.chunk, nl, p.long_rows.data_ptr<int32_t>(),
                                p.chunk_ptr.data_ptr<int64_t>(), lws.data_ptr(), lwb, st));
  if (nl >= 8)  // do the long rows lead the id range (degree-sorted order)?  one host read, at plan build only (ops.py)
    p.hub_first = p.long_rows.select(0, nl - 1).item<int32_t>() < 4 * nl;
  if (nl > 0)   // the serial hub walk starts its longest rows first (ggl_segplan.long_order; ops.py Engine._long_order)
    p.long_order = at::argsort(p.counts().index_select(0, p.long_rows.to(at::kLong)), /*stable=*/true, 0, /*descending=*/true).to(at::kInt);
}

static std::shared_ptr<SegPlan> build_plan(const Tensor &ids_in, int64_t N) {
  const auto dev = ids_in.device();
  const Api &a = api_for(dev);
  Tensor ids = ids_in.contiguous();
  void *st = stream_of(dev);
  auto p = std::make_shared<SegPlan>();
  p->N = N;
  p->E = ids.size(0);
  p->chunk = auto_chunk(a, p->E);
  auto i64 = ids.options().dtype(at::kLong);
  p->rowptr = at::empty({N + 1}, i64);
  Tensor perm = at::empty({std::max<int64_t>(p->E, 1)}, i64.dtype(at::kInt));
  const size_t wsb = a.ggl_plan_workspace_bytes(p->E, N);
  Tensor ws = at::empty({static_cast<int64_t>(wsb)}, i64.dtype(at::kByte));
  int32_t is_sorted = 0;
  int64_t max_len = 0;
  check(a, a.ggl_plan_build(ids.data_ptr<int64_t>(), p->E, N, perm.data_ptr<int32_t>(), p->rowptr.data_ptr<int64_t>(),
                            ws.data_ptr(), wsb, st, &is_sorted, &max_len));
  p->sorted = is_sorted != 0;
  p->max_len = max_len;
  if (!p->sorted) p->perm = perm.slice(0, 0, p->E);
  fill_long_rows(a, *p, st);
  p->uid = ++g_plans_built;
  return p;
}

// LRU keyed on the identity + version of a tensor; an entry dies with the tensor's storage (ops.py _PlanCache)
struct TensorKey {
  const void *storage = nullptr;
  int64_t offset = 0, numel = 0, version = 0, a = 0, b = 0;
  std::vector<int64_t> shape, stride;
  c10::Device dev = c10::Device(c10::kCPU);
  bool operator==(const TensorKey &o) const {
    return storage == o.storage && offset == o.offset && numel == o.numel && version == o.version && a == o.a &&
           b == o.b && dev == o.dev && shape == o.shape && stride == o.stride;
  }
  static TensorKey of(const Tensor &t, int64_t a, int64_t b) {
    TensorKey k;
    k.storage = t.storage().unsafeGetStorageImpl();
    k.offset = t.storage_offset();
    k.numel = t.numel();
    k.version = t.is_inference() ? 0 : static_cast<int64_t>(t._version());
    k.a = a;
    k.b = b;
    k.shape = t.sizes().vec();
    k.stride = t.strides().vec();
    k.dev = t.device();
    return k;
  }
};

template <typename V>
class Cache {
 public:
  explicit Cache(size_t cap) : cap_(cap) {}
  std::shared_ptr<V> get(const TensorKey &k) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto it = items_.begin(); it != items_.end(); ++it) {
      if (!(it->key == k)) continue;
      if (it->ref.expired()) {   // the storage is gone and its address was recycled: not the same tensor
        items_.erase(it);
        return nullptr;
      }
      items_.splice(items_.begin(), items_, it);
      return items_.front().val;
    }
    return nullptr;
  }
  void put(const Tensor &t, const TensorKey &k, std::shared_ptr<V> v) {
    std::lock_guard<std::mutex> g(mu_);
    items_.remove_if([&](const Item &i) { return i.key == k || i.ref.expired(); });
    items_.push_front(Item{k, t.storage().getWeakStorageImpl(), std::move(v)});
    while (items_.size() > cap_) items_.pop_back();
  }
  void clear() {
    std::lock_guard<std::mutex> g(mu_);
    items_.clear();
  }

 private:
  struct Item {
    TensorKey key;
    c10::weak_intrusive_ptr<c10::StorageImpl> ref;
    std::shared_ptr<V> val;
  };
  size_t cap_;
  std::mutex mu_;
  std::list<Item> items_;
};

static Cache<SegPlan> &seg_cache() {
  static Cache<SegPlan> c(16);
  return c;
}

static std::shared_ptr<SegPlan> seg_plan(const Tensor &ids, int64_t N) {
  TensorKey k = TensorKey::of(ids, N, 0);
  if (auto hit = seg_cache().get(k)) {
    ++g_plan_hits;
    return hit;
  }
  auto p = build_plan(ids, N);
  seg_cache().put(ids, k, p);
  return p;
}

static Tensor gather_i32(const Api &a, const Tensor &src_i64, const Tensor &perm) {
  Tensor src = src_i64.contiguous();
  Tensor out = at::empty({src.size(0)}, src.options().dtype(at::kInt));
  check(a, a.ggl_gather_i64_to_i32(src.data_ptr<int64_t>(), perm.defined() ? perm.data_ptr<int32_t>() : nullptr,
                                   src.size(0), out.data_ptr<int32_t>(), stream_of(src.device())));
  return out;
}

struct SortedW {   // edge weights in a plan's sorted order, once they have been seen twice
  int sightings = 0;
  Tensor sorted;
};

struct GraphPlan {
  int64_t N_dst = 0, N_src = 0, E = 0;
  std::shared_ptr<SegPlan> fwd, bwd;
  Tensor col, colT, rowidx;
  std::mutex mu;
  Cache<SortedW> weights{8};

  // share of the (sampled) edges whose endpoints lie within N / 64 ids of each other (ops.py GraphPlan.locality)
  double locality() const {
    const int64_t N = std::max(N_dst, N_src);
    if (E == 0 || N_dst != N_src) return 0.0;
    const int64_t S = std::min<int64_t>(int64_t(1) << 16, E);
    Tensor pos = at::arange(S, fwd->rowptr.options()) * (E / S);
    Tensor rows = at::searchsorted(fwd->rowptr, pos, /*out_int32=*/false, /*right=*/true) - 1;
    Tensor near = (col.index_select(0, pos).to(at::kLong) - rows).abs() < std::max<int64_t>(N / 64, 4096);
    return near.to(at::kFloat).mean().item<double>();
  }
  void schedule() {   // XCD runs where the node order carries locality (ops.py GraphPlan._schedule)
    const Api &a = api_for(fwd->rowptr.device());
    const bool need = a.ggl_policy_xcd_run_rows(E, 1.0) > 0 || a.ggl_policy_xcd_run_rows(E, 0.0) > 0;
    const int64_t run = a.ggl_policy_xcd_run_rows(E, need ? locality() : 0.0);   // (locality() is one host read)
    fwd->xcd_run = run;
    if (bwd) bwd->xcd_run = run;
  }
  void need_bwd(const Tensor &index) {   // CSC side, built on the first backward
    std::lock_guard<std::mutex> g(mu);
    if (bwd) return;
    const Api &a = api_for(index.device());
    auto b = seg_plan(index.select(0, 0), N_src);
    colT = gather_i32(a, index.select(0, 1), b->perm);
    b->xcd_run = fwd->xcd_run;
    bwd = b;
  }
  void need_rowidx(const Tensor &index) {
    std::lock_guard<std::mutex> g(mu);
    if (rowidx.defined()) return;
    rowidx = gather_i32(api_for(index.device()), index.select(0, 1), fwd->perm);
  }
  // transposed sorted position -> forward sorted position (int32 [E]; ops.py GraphPlan.posT)
  Tensor posT;
  void need_posT(const Tensor &index) {
    need_bwd(index);
    std::lock_guard<std::mutex> g(mu);
    if (posT.defined()) return;
    auto i32 = fwd->rowptr.options().dtype(at::kInt);
    Tensor ar = at::arange(E, i32);
    Tensor pf = fwd->perm.defined() ? fwd->perm : ar, pt = bwd->perm.defined() ? bwd->perm : ar;
    Tensor inv = at::empty({E}, i32);
    inv.index_put_({pf.to(at::kLong)}, ar);
    posT = inv.index_select(0, pt.to(at::kLong)).contiguous();
  }
  // forward sorted position -> transposed sorted position (int32 [E]; ops.py GraphPlan.tpos): where ggl_spmm_max_mask
  // scatters an edge's winner bits for the max backward's transposed walk
  Tensor tpos;
  void need_tpos(const Tensor &index) {
    need_posT(index);
    std::lock_guard<std::mutex> g(mu);
    if (tpos.defined()) return;
    const Api &a = api_for(posT.device());
    Tensor t = at::empty_like(posT);
    check(a, a.ggl_invert_perm(posT.data_ptr<int32_t>(), E, t.data_ptr<int32_t>(), stream_of(posT.device())));
    tpos = t;
  }
};

static Cache<GraphPlan> &graph_cache() {
  static Cache<GraphPlan> c(16);
  return c;
}

static void check_range(const Tensor &ids, int64_t n) {   // one host read per plan
  if (ids.numel() == 0) return;
  auto mm = at::aminmax(ids);
  Tensor both = at::stack({std::get<0>(mm), std::get<1>(mm)}).cpu();   // one host read
  TORCH_CHECK_INDEX(both[0].item<int64_t>() >= 0 && both[1].item<int64_t>() < n, "node id out of range [0, ", n, ")");
}

static std::shared_ptr<GraphPlan> graph_plan(const Tensor &index, int64_t n_dst, int64_t n_src) {
  TORCH_CHECK_INDEX(index.dim() == 2 && index.size(0) == 2, "edge index must have shape [2, E]");
  TORCH_CHECK(index.scalar_type() == at::kLong, "expected scalar type Long but found ", index.scalar_type());
  TensorKey k = TensorKey::of(index, n_dst, n_src);
  if (auto hit = graph_cache().get(k)) return hit;
  const Api &a = api_for(index.device());
  auto gp = std::make_shared<GraphPlan>();
  gp->N_dst = n_dst;
  gp->N_src = n_src;
  gp->E = index.size(1);
  gp->fwd = seg_plan(index.select(0, 1), n_dst);   // shared with segment ops on edge_index[1]
  check_range(index.select(0, 0), n_src);
  gp->col = gather_i32(a, index.select(0, 0), gp->fwd->perm);
  gp->schedule();
  graph_cache().put(index, k, gp);
  return gp;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward launches
// ---------------------------------------------------------------------------------------------------------------
static Tensor partial_for(const Api &a, const SegPlan &p, const Tensor &like, int64_t K, bool with_arg) {
  if (p.n_long == 0) return Tensor();
  const size_t nb = a.ggl_partial_bytes(dtype_code(like), p.n_chunks, K, with_arg ? 1 : 0);
  return at::empty({static_cast<int64_t>(nb) + 16}, like.options().dtype(at::kByte));
}

static std::vector<int64_t> out_shape(const Tensor &x, int64_t N) {
  auto s = x.sizes().vec();
  s[0] = N;
  return s;
}

enum class Red { Sum, Mean, Max };

static std::pair<Tensor, Tensor> segment_fwd(Red op, const Tensor &x, const SegPlan &p) {
  const auto dev = x.device();
  const Api &a = api_for(dev);
  const int64_t E = x.size(0);
  TORCH_CHECK_INDEX(E == p.E, "fisrt dimension of x and index should be same");   // segment_sum_cpu.cpp:17-19
  int64_t K = 1;
  for (int64_t d = 1; d < x.dim(); ++d) K *= x.size(d);
  Tensor out = at::empty(out_shape(x, p.N), x.options());
  void *st = stream_of(dev);
  const int code = dtype_code(x);
  const bool half = x.scalar_type() == at::kHalf || x.scalar_type() == at::kBFloat16;
  // f16 / bf16 sums accumulate in the storage type: chunk partials would not reproduce the serial result
  const bool unsplit = op != Red::Max && half;
  const bool hubs = unsplit && p.n_long > 0 && a.ggl_segment_hub16_supported(code, K, x.data_ptr(), out.data_ptr()) != 0;
  Tensor part = unsplit ? Tensor() : partial_for(a, p, x, K, op == Red::Max);
  ggl_segplan_t cs = p.c(part, unsplit && !hubs, hubs);
  if (op == Red::Max) {
    Tensor arg = at::empty(out_shape(x, p.N), x.options().dtype(at::kLong));
    check(a, a.ggl_segment_max(code, x.data_ptr(), &cs, K, out.data_ptr(), arg.data_ptr<int64_t>(), E, st));
    return {out, arg};
  }
  check(a, (op == Red::Sum ? a.ggl_segment_sum : a.ggl_segment_mean)(code, x.data_ptr(), &cs, K, out.data_ptr(), st));
  if (hubs) {
    ggl_segplan_t full = p.c(Tensor());
    check(a, a.ggl_segment_hub16(code, op == Red::Mean ? 1 : 0, x.data_ptr(), &full, K, out.data_ptr(), st));
  }
  return {out, Tensor()};
}

static std::pair<const float *, int> weights_for(const Api &a, GraphPlan &gp, const SegPlan &p, const Tensor &w,
                                                 Tensor &keep) {
  if (!w.defined()) return {nullptr, 0};
  if (!p.perm.defined()) return {w.data_ptr<float>(), 0};
  TensorKey k = TensorKey::of(w, static_cast<int64_t>(p.uid), 0);
  auto hit = gp.weights.get(k);
  if (!hit) {   // first sight: remember it, the kernel reads w[perm[p]] itself
    auto s = std::make_shared<SortedW>();
    s->sightings = 1;
    gp.weights.put(w, k, s);
    return {w.data_ptr<float>(), 0};
  }
  if (!hit->sorted.defined()) {
    const int64_t H = p.E > 0 ? std::max<int64_t>(w.numel() / p.E, 1) : 1;
    Tensor ws = at::empty_like(w);
    check(a, a.ggl_gather_rows_f32(w.data_ptr<float>(), p.perm.data_ptr<int32_t>(), p.E, H, ws.data_ptr<float>(),
                                   stream_of(w.device())));
    hit->sorted = ws;
  }
  keep = hit->sorted;
  return {keep.data_ptr<float>(), 1};
}

enum class SpOp { Sum, Mean, Max, MeanBwd, MaxBwd };

static std::pair<Tensor, Tensor> spmm_fwd(SpOp op, GraphPlan &gp, const SegPlan &p, const Tensor &col, const Tensor &w,
                                          const Tensor &x, int64_t n_out, const Tensor &aux = Tensor(),
                                          const Tensor &tpos = Tensor()) {
  const auto dev = x.device();
  const Api &a = api_for(dev);
  int64_t K = 1;
  for (int64_t d = 1; d < x.dim(); ++d) K *= x.size(d);
  if ((op == SpOp::Sum || op == SpOp::Mean || op == SpOp::Max) && x.dim() == 2) {
    // one zero-padded copy where the kernels want whole cache lines / 16-byte rows (ggl_policy_spmm_width; ops.py _spmm_fwd)
    const int64_t Kp = a.ggl_policy_spmm_width(op == SpOp::Max ? 1 : 0, K, p.E, x.size(0));
    if (Kp != K) {
      Tensor xp = at::constant_pad_nd(x, {0, Kp - K});
      auto r = spmm_fwd(op, gp, p, col, w, xp, n_out, aux);
      return {r.first.slice(1, 0, K).contiguous(), r.second.defined() ? r.second.slice(1, 0, K).contiguous() : Tensor()};
    }
  }
  Tensor out = at::empty(out_shape(x, n_out), x.options());
  void *st = stream_of(dev);
  Tensor part = partial_for(a, p, x, K, op == SpOp::Max);
  ggl_segplan_t cs = p.c(part);
  Tensor keep;
  auto [wp, by_pos] = weights_for(a, gp, p, w, keep);
  const int32_t *c = col.data_ptr<int32_t>();
  const float *xp = x.data_ptr<float>();
  float *op_ = out.data_ptr<float>();
  switch (op) {
    case SpOp::Sum: check(a, a.ggl_spmm_sum(&cs, c, wp, by_pos, xp, K, op_, st)); break;
    case SpOp::Mean: check(a, a.ggl_spmm_mean(&cs, c, wp, by_pos, xp, K, op_, st)); break;
    case SpOp::Max: {
      Tensor arg = at::empty(out.sizes(), out.options().dtype(at::kLong));
      check(a, a.ggl_spmm_max(&cs, c, wp, by_pos, xp, K, op_, arg.data_ptr<int64_t>(), st));
      return {out, arg};
    }
    case SpOp::MeanBwd:
      if (x.dim() == 2 && a.ggl_policy_mean_bwd_prescale(p.E, x.size(0))) {
        // the division depends on the destination row only: once per row, then the plain transposed SpMM-sum
        Tensor cnt = (aux.slice(0, 1) - aux.slice(0, 0, aux.size(0) - 1)).clamp_min(1).to(at::kFloat).unsqueeze(1);
        Tensor xs = x / cnt;
        check(a, a.ggl_spmm_sum(&cs, c, wp, by_pos, xs.data_ptr<float>(), K, op_, st));
      } else {
        check(a, a.ggl_spmm_mean_bwd(&cs, c, wp, by_pos, xp, aux.data_ptr<int64_t>(), K, op_, st));
      }
      break;
    case SpOp::MaxBwd: {
      // the form is the library's decision (ggl_policy_maxbwd_form: the winner mask only where its E x K/8-byte transient
      // pays); `tpos` is defined iff the caller found the mask form chosen.  An allocation failure falls back to the int32 copy.
      Tensor mask;
      if (tpos.defined()) {
        try {
          mask = at::empty({static_cast<int64_t>(a.ggl_spmm_max_mask_bytes(p.E, K) / 4) + 4}, x.options().dtype(at::kInt));
        } catch (const c10::Error &) {
          mask = Tensor();
        }
      }
      if (mask.defined()) {
        // a 1-bit winner mask built in destination order; records in forward order, read at posT[t] (ops.py _spmm_fwd) —
        // `tpos` here IS posT unless the A/B knob maxbwd_mask_scatter asks for records scattered to transposed positions
        const bool scatter = a.ggl_get_option("maxbwd_mask_scatter") != 0;
        ggl_segplan_t fs = gp.fwd->c(Tensor());
        check(a, a.ggl_spmm_max_mask(&fs, gp.col.data_ptr<int32_t>(), scatter ? tpos.data_ptr<int32_t>() : nullptr,
                                     aux.data_ptr<int64_t>(), K, reinterpret_cast<uint32_t *>(mask.data_ptr<int32_t>()), st));
        check(a, a.ggl_spmm_max_bwd_mask(&cs, c, wp, by_pos, xp, reinterpret_cast<const uint32_t *>(mask.data_ptr<int32_t>()),
                                         scatter ? nullptr : tpos.data_ptr<int32_t>(), K, op_, st));
      } else if (tpos.defined() || a.ggl_get_option("maxbwd_arg32") != 0) {   // witnesses from a compact int32 copy (one [N, K] pass)
        Tensor aux32 = aux.to(at::kInt);
        check(a, a.ggl_spmm_max_bwd32(&cs, c, wp, by_pos, xp, aux32.data_ptr<int32_t>(), K, op_, st));
      } else {
        check(a, a.ggl_spmm_max_bwd(&cs, c, wp, by_pos, xp, aux.data_ptr<int64_t>(), K, op_, st));
      }
      break;
    }
  }
  return {out, Tensor()};
}

static Tensor bspmm_fwd(GraphPlan &gp, const SegPlan &p, const Tensor &col, const Tensor &w, const Tensor &x,
                        int64_t n_out) {
  const Api &a = api_for(x.device());
  const int64_t H = x.size(1), C = x.size(2);
  Tensor out = at::empty({n_out, H, C}, x.options());
  Tensor part = partial_for(a, p, x, H * C, false);
  ggl_segplan_t cs = p.c(part);
  Tensor keep;
  auto [wp, by_pos] = weights_for(a, gp, p, w, keep);
  check(a, a.ggl_bspmm_sum(&cs, col.data_ptr<int32_t>(), wp, by_pos, x.data_ptr<float>(), H, C, out.data_ptr<float>(),
                           stream_of(x.device())));
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Argument checks (the reference's predicates and exception types: segment_sum_cpu.cpp:13-19, spmm_sum_cpu.cpp:22)
// ---------------------------------------------------------------------------------------------------------------
static void seg_args(const Tensor &x, const Tensor &index) {
  same_device({&x, &index});
  TORCH_CHECK_INDEX(index.dim() == 1, "index dimension should be 1, but got ", index.dim());
  TORCH_CHECK_INDEX(x.dim() >= 1 && x.size(0) == index.size(0), "fisrt dimension of x and index should be same");
  TORCH_CHECK(index.scalar_type() == at::kLong, "expected scalar type Long but found ", index.scalar_type());
}

static void f32(const char *name, const Tensor &t) {
  TORCH_CHECK(t.scalar_type() == at::kFloat, "expected scalar type Float but found ", t.scalar_type(), " (", name, ")");
}

using OptT_ = c10::optional<Tensor>;
static Tensor opt(const c10::optional<Tensor> &t) { return t.has_value() ? *t : Tensor(); }
// the kernels read weight.data_ptr() as E dense floats (spmm_sum_cpu.cpp:48-50 makes it contiguous in backward too)
static Tensor opt_dense(const c10::optional<Tensor> &t) { return t.has_value() && t->defined() ? t->contiguous() : Tensor(); }

// ---------------------------------------------------------------------------------------------------------------
// Backend kernels (forward only; these are what the CUDA / CPU keys run, e.g. under no_grad)
// ---------------------------------------------------------------------------------------------------------------
static Tensor segment_sum_kernel(const Tensor &x, const Tensor &index, int64_t N) {
  seg_args(x, index);
  c10::OptionalDeviceGuard guard(x.device());
  return segment_fwd(Red::Sum, x.contiguous(), *seg_plan(index, N)).first;
}
static Tensor segment_mean_kernel(const Tensor &x, const Tensor &index, int64_t N) {
  seg_args(x, index);
  c10::OptionalDeviceGuard guard(x.device());
  return segment_fwd(Red::Mean, x.contiguous(), *seg_plan(index, N)).first;
}
static std::tuple<Tensor, Tensor> segment_max_kernel(const Tensor &x, const Tensor &index, int64_t N) {
  seg_args(x, index);
  c10::OptionalDeviceGuard guard(x.device());
  auto r = segment_fwd(Red::Max, x.contiguous(), *seg_plan(index, N));
  return {r.first, r.second};
}

struct SpArgs {
  std::shared_ptr<GraphPlan> gp;
  Tensor w, x;
};
static SpArgs spmm_args(const Tensor &index, const c10::optional<Tensor> &weight, const Tensor &x) {
  Tensor w = opt(weight);
  same_device({&index, &w, &x});
  f32("x", x);
  if (w.defined()) f32("weight", w);
  TORCH_CHECK(x.dim() >= 1, "x must have a node dimension");
  SpArgs s;
  s.gp = graph_plan(index, x.size(0), x.size(0));   // gspmm.cpp:16 out = zeros_like(x): square
  if (w.defined()) TORCH_CHECK(w.numel() % std::max<int64_t>(s.gp->E, 1) == 0 && w.size(0) == s.gp->E,
                               "edge weight must hold one row per edge: got ", w.sizes(), " for ", s.gp->E, " edges");
  s.w = w.defined() ? w.contiguous() : w;
  s.x = x.contiguous();
  return s;
}

static Tensor spmm_kernel(SpOp op, const Tensor &index, const c10::optional<Tensor> &weight, const Tensor &x,
                          Tensor *arg = nullptr) {
  c10::OptionalDeviceGuard guard(x.device());
  SpArgs s = spmm_args(index, weight, x);
  auto r = spmm_fwd(op, *s.gp, *s.gp->fwd, s.gp->col, s.w, s.x, s.gp->N_dst);
  if (arg != nullptr) *arg = r.second;
  return r.first;
}
static Tensor spmm_sum_kernel(const Tensor &i, const c10::optional<Tensor> &w, const Tensor &x) {
  return spmm_kernel(SpOp::Sum, i, w, x);
}
static Tensor spmm_mean_kernel(const Tensor &i, const c10::optional<Tensor> &w, const Tensor &x) {
  return spmm_kernel(SpOp::Mean, i, w, x);
}
static Tensor spmm_max_kernel(const Tensor &i, const c10::optional<Tensor> &w, const Tensor &x) {
  return spmm_kernel(SpOp::Max, i, w, x);
}

static int64_t head_pad(const GraphPlan &gp, const Tensor &x) {   // channels to append per head (ggl_policy_head_channels)
  return api_for(x.device()).ggl_policy_head_channels(x.size(2), gp.E, x.size(0)) - x.size(2);
}
static void bspmm_check(const Tensor &w, const Tensor &x) {
  TORCH_CHECK(x.dim() == 3, "bspmm expects x of shape [num_nodes, heads, channels]");
  TORCH_CHECK(w.defined() && w.dim() == 2 && w.size(1) == x.size(1), "bspmm expects weight of shape [num_edges, heads]");
}
static Tensor bspmm_sum_kernel(const Tensor &index, const Tensor &weight, const Tensor &x) {
  c10::OptionalDeviceGuard guard(x.device());
  TORCH_CHECK(x.dim() == 3, "bspmm expects x of shape [num_nodes, heads, channels]");
  SpArgs s = spmm_args(index, weight, x);
  bspmm_check(s.w, s.x);
  const int64_t C = s.x.size(2);
  const int64_t hp = head_pad(*s.gp, s.x);
  Tensor xin = hp > 0 ? at::constant_pad_nd(s.x, {0, hp}) : s.x;
  Tensor out = bspmm_fwd(*s.gp, *s.gp->fwd, s.gp->col, s.w, xin, s.gp->N_dst);
  return xin.size(2) == C ? out : out.slice(2, 0, C).contiguous();
}

// ---------------------------------------------------------------------------------------------------------------
// Backward passes as dispatcher ops of their own (backend + Meta kernels): the autograd formulas below only CALL ops,
// so a graph that contains them traces end to end under FakeTensor / torch.compile (AOTAutograd), forward and backward
// ---------------------------------------------------------------------------------------------------------------
static int64_t width_of(c10::IntArrayRef shape) {
  int64_t K = 1;
  for (size_t d = 1; d < shape.size(); ++d) K *= shape[d];
  return K;
}

// gin[e, :] = gout[ids[e], :]                                            (segment_sum.cpp:43-54)
static Tensor segment_sum_backward_kernel(const Tensor &grad, const Tensor &index, c10::IntArrayRef x_shape) {
  Tensor g = grad.contiguous();
  c10::OptionalDeviceGuard guard(g.device());
  const Api &a = api_for(g.device());
  Tensor gin = at::empty(x_shape, g.options());
  Tensor ids = index.contiguous();
  check(a, a.ggl_segment_sum_bwd(dtype_code(g), g.data_ptr(), ids.data_ptr<int64_t>(), x_shape[0], width_of(x_shape),
                                 gin.data_ptr(), stream_of(g.device())));
  return gin;
}
// gin[e, :] = gout[ids[e], :] / count[ids[e]]                             (segment_mean.cpp:44-63)
static Tensor segment_mean_backward_kernel(const Tensor &grad, const Tensor &index, int64_t N, c10::IntArrayRef x_shape) {
  Tensor g = grad.contiguous();
  TORCH_CHECK(g.is_floating_point(), "segment_mean backward needs a floating dtype");
  c10::OptionalDeviceGuard guard(g.device());
  const Api &a = api_for(g.device());
  Tensor gin = at::empty(x_shape, g.options());
  Tensor ids = index.contiguous();
  auto plan = seg_plan(index, N);    // (the forward's plan: a cache hit)
  check(a, a.ggl_segment_mean_bwd(dtype_code(g), g.data_ptr(), ids.data_ptr<int64_t>(), plan->rowptr.data_ptr<int64_t>(),
                                  x_shape[0], width_of(x_shape), gin.data_ptr(), stream_of(g.device())));
  return gin;
}
// gin = 0; gin[arg[s, k], k] = gout[s, k] where the segment is not empty   (segment_max.cpp:48-61)
static Tensor segment_max_backward_kernel(const Tensor &grad, const Tensor &arg, c10::IntArrayRef x_shape) {
  Tensor g = grad.contiguous();
  c10::OptionalDeviceGuard guard(g.device());
  const Api &a = api_for(g.device());
  Tensor gin = at::empty(x_shape, g.options());
  Tensor ar = arg.contiguous();
  check(a, a.ggl_segment_max_bwd(dtype_code(g), g.data_ptr(), ar.data_ptr<int64_t>(), x_shape[0], ar.size(0),
                                 width_of(x_shape), gin.data_ptr(), stream_of(g.device())));
  return gin;
}

static std::shared_ptr<GraphPlan> bwd_plan(const Tensor &index, int64_t n) {
  auto gp = graph_plan(index, n, n);
  gp->need_bwd(index);
  return gp;
}
// gx[src] += w[e] * g[dst]: the same walk on the transposed plan          (spmm_sum_cpu.cpp:43-80)
static Tensor spmm_sum_backward_kernel(const Tensor &index, const c10::optional<Tensor> &weight, const Tensor &grad) {
  Tensor g = grad.contiguous(), w = opt_dense(weight);
  c10::OptionalDeviceGuard guard(g.device());
  auto gp = bwd_plan(index, g.size(0));
  return spmm_fwd(SpOp::Sum, *gp, *gp->bwd, gp->colT, w, g, gp->N_src).first;
}
static Tensor spmm_mean_backward_kernel(const Tensor &index, const c10::optional<Tensor> &weight, const Tensor &grad) {
  Tensor g = grad.contiguous(), w = opt_dense(weight);
  c10::OptionalDeviceGuard guard(g.device());
  auto gp = bwd_plan(index, g.size(0));
  return spmm_fwd(SpOp::MeanBwd, *gp, *gp->bwd, gp->colT, w, g, gp->N_src, gp->fwd->rowptr).first;
}
static Tensor spmm_max_backward_kernel(const Tensor &index, const c10::optional<Tensor> &weight, const Tensor &grad,
                                       const Tensor &arg) {
  Tensor g = grad.contiguous(), w = opt_dense(weight);
  c10::OptionalDeviceGuard guard(g.device());
  auto gp = bwd_plan(index, g.size(0));
  Tensor tpos;
  const int64_t Kw = g.dim() >= 2 ? g.numel() / std::max<int64_t>(g.size(0), 1) : 1;
  // (the inverse-permutation passes and the cached int32[E] only where the mask form is the one chosen for this K)
  if (api_for(g.device()).ggl_policy_maxbwd_form(gp->E, g.size(0), Kw) == 2) {
    if (api_for(g.device()).ggl_get_option("maxbwd_mask_scatter") != 0) {
      gp->need_tpos(index.contiguous());
      tpos = gp->tpos;
    } else {
      gp->need_posT(index.contiguous());
      tpos = gp->posT;
    }
  }
  return spmm_fwd(SpOp::MaxBwd, *gp, *gp->bwd, gp->colT, w, g, gp->N_src, arg.contiguous(), tpos).first;
}
static std::tuple<Tensor, Tensor> spmm_max_arg_kernel(const Tensor &i, const c10::optional<Tensor> &w, const Tensor &x) {
  Tensor arg;
  Tensor out = spmm_kernel(SpOp::Max, i, w, x, &arg);
  return {out, arg};
}
// (gw, gx): gx = the transposed walk; gw[e, h] = sum_c x[src, h, c] * g[dst, h, c]   (bspmm_sum_cpu.cpp:58-113)
static std::tuple<Tensor, Tensor> bspmm_sum_backward_kernel(const Tensor &index, const Tensor &weight, const Tensor &x_in,
                                                            const Tensor &grad) {
  c10::OptionalDeviceGuard guard(grad.device());
  const Api &a = api_for(grad.device());
  Tensor w = weight.contiguous();
  auto gp = bwd_plan(index, x_in.size(0));
  const int64_t C0 = x_in.size(2), H = x_in.size(1);
  const int64_t pad = head_pad(*gp, x_in);
  Tensor x = (pad > 0 ? at::constant_pad_nd(x_in, {0, pad}) : x_in).contiguous();
  Tensor g = (pad > 0 ? at::constant_pad_nd(grad, {0, pad}) : grad).contiguous();
  const int64_t C = C0 + pad;
  Tensor gx = bspmm_fwd(*gp, *gp->bwd, gp->colT, w, g, gp->N_src);
  Tensor gw = at::empty_like(w);
  void *st = stream_of(g.device());
  if (a.ggl_policy_gradw_sorted(H, C)) {   // along the destination-sorted plan, strips staged through LDS (edgedot.hip)
    gp->need_rowidx(index);
    const size_t sb = a.ggl_bspmm_grad_w_sorted_scratch_bytes(gp->E, gp->N_dst, H, C);
    Tensor scratch = sb > 0 ? at::empty({static_cast<int64_t>(sb / 4)}, g.options()) : Tensor();
    ggl_segplan_t cs = gp->fwd->c(Tensor());
    check(a, a.ggl_bspmm_grad_w_sorted(&cs, gp->col.data_ptr<int32_t>(), gp->rowidx.data_ptr<int32_t>(),
                                       x.data_ptr<float>(), g.data_ptr<float>(), H, C, gw.data_ptr<float>(),
                                       scratch.defined() ? scratch.data_ptr<float>() : nullptr, st));
  } else {
    Tensor idx = index.contiguous();
    check(a, a.ggl_bspmm_grad_w(idx.data_ptr<int64_t>(), x.data_ptr<float>(), g.data_ptr<float>(), gp->E, H, C,
                                gw.data_ptr<float>(), st));
  }
  if (pad > 0) gx = gx.slice(2, 0, C0).contiguous();
  return {gw, gx};
}

// ---------------------------------------------------------------------------------------------------------------
// The fused route: edge-softmax + aggregate, the layer epilogue in the aggregate's
// store, one static-shape sampler hop — registered like the seven operators above: forward and backward passes are
// dispatcher ops of their own, the autograd formulas only call ops.  (Until round 4 these existed as Python-registered
// ops only, gammagl_amd/torch_ops.py: 20-30 us of Python + ctypes per call.)
// ---------------------------------------------------------------------------------------------------------------
// device-resident {seed, offset} of the fused dropouts, seeded from torch's CPU generator on first use (follows
// torch.manual_seed, like ops.py Engine._rng_state); ggl::reseed() forgets it
static std::mutex g_rng_mu;
static std::vector<std::pair<c10::Device, Tensor>> g_rng;
static Tensor rng_state(const c10::Device &d) {
  std::lock_guard<std::mutex> g(g_rng_mu);
  for (auto &e : g_rng)
    if (e.first == d) return e.second;
  const int64_t seed = at::randint(0, int64_t(1) << 62, {1}, at::TensorOptions().dtype(at::kLong)).item<int64_t>();
  Tensor st = at::tensor({seed, int64_t(0)}, at::TensorOptions().dtype(at::kLong)).to(d);
  g_rng.emplace_back(d, st);
  return st;
}
static void reseed() {
  std::lock_guard<std::mutex> g(g_rng_mu);
  g_rng.clear();
}

static void gat_check(const Tensor &el, const Tensor &er, const Tensor &x) {
  same_device({&el, &er, &x});
  f32("el", el); f32("er", er); f32("x", x);
  TORCH_CHECK(x.dim() == 3 && el.dim() == 2 && er.dim() == 2 && el.size(1) == x.size(1) && er.size(1) == x.size(1) &&
              el.size(0) == x.size(0), "gat_fused expects el [N_src, H], er [N_dst, H], x [N_src, H, C]");
}

// (out, rowmax, rowden, rng_used, fast): one launch (+ the hub-chunk combine inside the library)
static std::tuple<Tensor, Tensor, Tensor, Tensor, bool> gat_forward(GraphPlan &gp, const Tensor &el_, const Tensor &er_,
                                                                    const Tensor &x_, double slope, double p) {
  const auto dev = x_.device();
  const Api &a = api_for(dev);
  Tensor el = el_.contiguous(), er = er_.contiguous(), x = x_.contiguous();
  TORCH_CHECK(er.size(0) == gp.N_dst && x.size(0) == gp.N_src, "gat_fused: er has ", er.size(0), " rows for ", gp.N_dst,
              " destinations, x ", x.size(0), " for ", gp.N_src, " sources");
  const int64_t N = gp.N_dst, H = x.size(1), C = x.size(2);
  Tensor out = at::empty({N, H, C}, x.options()), rmax = at::empty({N, H}, x.options()), rden = at::empty({N, H}, x.options());
  Tensor part;
  if (gp.fwd->n_long > 0)
    part = at::empty({static_cast<int64_t>(a.ggl_gat_partial_bytes(gp.fwd->n_chunks, H, C)) + 16}, x.options().dtype(at::kByte));
  ggl_segplan_t cs = gp.fwd->c(part);
  Tensor rng, rng_used;
  if (p > 0) {
    rng = rng_state(dev);
    rng_used = rng.clone();   // the {seed, offset} this launch reads; the backward redraws the mask
  }
  static const bool want_fast = env_int("GGL_GAT_FAST", 1) != 0;
  const bool fast = want_fast && a.ggl_gat_fast_supported(H, C) != 0;
  void *st = stream_of(dev);
  int64_t *rp = rng.defined() ? rng.data_ptr<int64_t>() : nullptr;
  if (fast)
    check(a, a.ggl_gat_fast_fwd(&cs, gp.col.data_ptr<int32_t>(), el.data_ptr<float>(), er.data_ptr<float>(),
                                x.data_ptr<float>(), x.size(0), static_cast<float>(slope), H, C, static_cast<float>(p), rp,
                                out.data_ptr<float>(), rmax.data_ptr<float>(), rden.data_ptr<float>(), st));
  else
    check(a, a.ggl_gat_fused_fwd(&cs, gp.col.data_ptr<int32_t>(), el.data_ptr<float>(), er.data_ptr<float>(),
                                 x.data_ptr<float>(), static_cast<float>(slope), H, C, static_cast<float>(p), rp,
                                 out.data_ptr<float>(), rmax.data_ptr<float>(), rden.data_ptr<float>(), st));
  return {out, rmax, rden, rng_used.defined() ? rng_used : at::empty({0}, x.options().dtype(at::kLong)), fast};
}

// (gel, ger, gx): the destination walk and the source walk
static std::tuple<Tensor, Tensor, Tensor> gat_backward(GraphPlan &gp, const Tensor &colT, const Tensor &posT,
                                                       const Tensor &el, const Tensor &er, const Tensor &x, const Tensor &g_,
                                                       const Tensor &out, const Tensor &rmax, const Tensor &rden,
                                                       const Tensor &rng_used, double slope, double p, bool fast) {
  const auto dev = x.device();
  const Api &a = api_for(dev);
  Tensor g = g_.contiguous();
  const int64_t H = x.size(1), C = x.size(2);
  void *st = stream_of(dev);
  const SegPlan &fw = *gp.fwd, &bw = *gp.bwd;
  Tensor ger = at::empty_like(er), gx = at::empty({gp.N_src, H, C}, x.options()), gel = at::empty({gp.N_src, H}, x.options());
  // (the fast destination walk keeps four DOUBLE sums per chunk and head: 8 H floats' worth per chunk, include/ggl_mpops.h)
  Tensor part_f = partial_for(a, fw, x, fast ? 8 * H : H, false), part_t = partial_for(a, bw, x, H * C + H, false);
  ggl_segplan_t cs = fw.c(part_f), csT = bw.c(part_t);
  const int64_t *ru = (p > 0 && rng_used.numel() == 2) ? rng_used.data_ptr<int64_t>() : nullptr;
  if (fast) {
    Tensor stats = at::empty({gp.N_dst, H, 4}, x.options());
    check(a, a.ggl_gat_fast_bwd(&cs, gp.col.data_ptr<int32_t>(), &csT, colT.data_ptr<int32_t>(),
                                p > 0 ? posT.data_ptr<int32_t>() : nullptr, el.data_ptr<float>(), er.data_ptr<float>(),
                                x.data_ptr<float>(), g.data_ptr<float>(), out.data_ptr<float>(), rmax.data_ptr<float>(),
                                rden.data_ptr<float>(), static_cast<float>(slope), H, C, static_cast<float>(p), ru,
                                stats.data_ptr<float>(), gx.data_ptr<float>(), gel.data_ptr<float>(), ger.data_ptr<float>(), st));
    return {gel, ger, gx};
  }
  // alpha and de interleaved [E, H, 2]: the source-side walk fetches both with one 64-byte line
  Tensor ad = at::empty({std::max<int64_t>(gp.E, 1), H, 2}, x.options());
  float *alpha = ad.data_ptr<float>(), *de = alpha + 1;
  check(a, a.ggl_gat_fused_bwd_dst(&cs, gp.col.data_ptr<int32_t>(), nullptr, el.data_ptr<float>(), er.data_ptr<float>(),
                                   x.data_ptr<float>(), g.data_ptr<float>(), out.data_ptr<float>(), rmax.data_ptr<float>(),
                                   rden.data_ptr<float>(), static_cast<float>(slope), H, C, static_cast<float>(p), ru, alpha,
                                   de, ger.data_ptr<float>(), nullptr, st));
  check(a, a.ggl_gat_fused_bwd_src(&csT, colT.data_ptr<int32_t>(), posT.data_ptr<int32_t>(), alpha, de, g.data_ptr<float>(),
                                   H, C, gx.data_ptr<float>(), gel.data_ptr<float>(), st));
  return {gel, ger, gx};
}

static std::shared_ptr<GraphPlan> gat_plan(const Tensor &index, int64_t n_dst, int64_t n_src) {
  TORCH_CHECK(index.scalar_type() == at::kLong, "expected scalar type Long but found ", index.scalar_type());
  return graph_plan(index.contiguous(), n_dst, n_src);
}

// out[i, h, :] = sum_{j -> i} dropout(softmax_i(LeakyReLU(el[j, h] + er[i, h]))) x[j, h, :]   (gat_conv.py:103-112)
static std::tuple<Tensor, Tensor, Tensor, Tensor, bool> gat_fused_forward_kernel(const Tensor &index, const Tensor &el,
                                                                                const Tensor &er, const Tensor &x,
                                                                                double slope, int64_t num_nodes, double p) {
  gat_check(el, er, x);
  TORCH_CHECK(p >= 0.0 && p < 1.0, "dropout_rate must be in [0, 1)");
  c10::OptionalDeviceGuard guard(x.device());
  auto gp = gat_plan(index, num_nodes, x.size(0));
  return gat_forward(*gp, el, er, x, slope, p);
}
static std::tuple<Tensor, Tensor, Tensor> gat_fused_backward_kernel(const Tensor &index, const Tensor &el, const Tensor &er,
                                                                    const Tensor &x, const Tensor &grad, const Tensor &out,
                                                                    const Tensor &rmax, const Tensor &rden,
                                                                    const Tensor &rng_used, double slope, int64_t num_nodes,
                                                                    double p, bool fast) {
  c10::OptionalDeviceGuard guard(x.device());
  Tensor idx = index.contiguous();
  auto gp = gat_plan(idx, num_nodes, x.size(0));
  gp->need_bwd(idx);
  if (p > 0 || !fast) gp->need_posT(idx);
  return gat_backward(*gp, gp->colT, gp->posT, el.contiguous(), er.contiguous(), x.contiguous(), grad, out, rmax, rden,
                      rng_used, slope, p, fast);
}

// The same op over a CSR the caller already holds — the argument list of dgNN's GATConvFuse
// (layers/conv/fusedgat_conv.py:121): rows of the CSR aggregate, no sort.  The plan is cached on row_ptr.
struct CsrExtra {   // what besides row_ptr identifies the five-tensor structure
  const void *p[4];
  int64_t v[4];
  bool operator==(const CsrExtra &o) const { return std::memcmp(this, &o, sizeof(CsrExtra)) == 0; }
};
static Cache<GraphPlan> &csr_cache() {
  static Cache<GraphPlan> c(8);
  return c;
}
static std::mutex g_csr_mu;
static std::list<std::pair<std::weak_ptr<GraphPlan>, CsrExtra>> g_csr_extra;

static Tensor own_i32(const Tensor &t) { return t.scalar_type() == at::kInt ? t.clone().contiguous() : t.to(at::kInt).contiguous(); }
static std::shared_ptr<SegPlan> plan_from_rowptr(const Tensor &rowptr, int64_t E) {
  const Api &a = api_for(rowptr.device());
  auto p = std::make_shared<SegPlan>();
  p->N = rowptr.size(0) - 1;
  p->E = E;
  p->chunk = auto_chunk(a, E);
  p->rowptr = rowptr.to(at::kLong).contiguous();
  if (p->rowptr.data_ptr() == rowptr.data_ptr()) p->rowptr = p->rowptr.clone();   // the plan keeps its OWN copy
  p->sorted = true;
  p->max_len = p->N > 0 ? p->counts().max().item<int64_t>() : 0;
  fill_long_rows(a, *p, stream_of(rowptr.device()));
  p->uid = ++g_plans_built;
  return p;
}
static std::shared_ptr<GraphPlan> csr_plan(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &col_ptr,
                                           const Tensor &row_ind, const Tensor &permute) {
  same_device({&row_ptr, &col_ind, &col_ptr, &row_ind, &permute});
  const int64_t n_rows = row_ptr.size(0) - 1, n_cols = col_ptr.size(0) - 1, E = col_ind.size(0);
  for (const Tensor *t : {&row_ptr, &col_ind, &col_ptr, &row_ind, &permute})
    TORCH_CHECK(t->dim() == 1 && (t->scalar_type() == at::kInt || t->scalar_type() == at::kLong),
                "GATConvFuse: the CSR / CSC tensors must be 1-D int32 / int64");
  TORCH_CHECK(row_ind.size(0) == E && permute.size(0) == E, "GATConvFuse: col_ind, row_ind and permute must have one entry per edge");
  TensorKey k = TensorKey::of(row_ptr, n_rows, n_cols);
  CsrExtra ex{};
  const Tensor *others[4] = {&col_ind, &col_ptr, &row_ind, &permute};
  for (int i = 0; i < 4; ++i) {
    ex.p[i] = others[i]->data_ptr();
    ex.v[i] = static_cast<int64_t>(others[i]->_version());
  }
  if (auto hit = csr_cache().get(k)) {
    std::lock_guard<std::mutex> g(g_csr_mu);
    for (auto &e : g_csr_extra)
      if (e.first.lock() == hit && e.second == ex) return hit;
  }
  check_range(col_ind, n_cols);
  check_range(row_ind, n_rows);
  check_range(permute, std::max<int64_t>(E, 1));
  for (const Tensor *ptr : {&row_ptr, &col_ptr}) {   // one-off (per plan) host reads
    Tensor p64 = ptr->to(at::kLong);
    TORCH_CHECK(p64[0].item<int64_t>() == 0 && p64[-1].item<int64_t>() == E &&
                    (p64.size(0) < 2 || !(p64.slice(0, 1) < p64.slice(0, 0, p64.size(0) - 1)).any().item<bool>()),
                "GATConvFuse: a row pointer must rise from 0 to the number of edges (", E, ")");
  }
  auto gp = std::make_shared<GraphPlan>();
  gp->N_dst = n_rows;
  gp->N_src = n_cols;
  gp->E = E;
  gp->fwd = plan_from_rowptr(row_ptr, E);
  gp->col = own_i32(col_ind);
  gp->bwd = plan_from_rowptr(col_ptr, E);
  gp->colT = own_i32(row_ind);
  gp->posT = own_i32(permute);
  gp->schedule();
  csr_cache().put(row_ptr, k, gp);
  {
    std::lock_guard<std::mutex> g(g_csr_mu);
    g_csr_extra.remove_if([](const std::pair<std::weak_ptr<GraphPlan>, CsrExtra> &e) { return e.first.expired(); });
    g_csr_extra.emplace_back(gp, ex);
  }
  return gp;
}
static std::tuple<Tensor, Tensor, Tensor, Tensor, bool> gat_csr_forward_kernel(const Tensor &row_ptr, const Tensor &col_ind,
                                                                              const Tensor &col_ptr, const Tensor &row_ind,
                                                                              const Tensor &permute, const Tensor &el,
                                                                              const Tensor &er, const Tensor &x, double slope,
                                                                              double p) {
  gat_check(el, er, x);
  TORCH_CHECK(p >= 0.0 && p < 1.0, "dropout_rate must be in [0, 1)");
  c10::OptionalDeviceGuard guard(x.device());
  auto gp = csr_plan(row_ptr, col_ind, col_ptr, row_ind, permute);
  return gat_forward(*gp, el, er, x, slope, p);
}
static std::tuple<Tensor, Tensor, Tensor> gat_csr_backward_kernel(const Tensor &row_ptr, const Tensor &col_ind,
                                                                  const Tensor &col_ptr, const Tensor &row_ind,
                                                                  const Tensor &permute, const Tensor &el, const Tensor &er,
                                                                  const Tensor &x, const Tensor &grad, const Tensor &out,
                                                                  const Tensor &rmax, const Tensor &rden,
                                                                  const Tensor &rng_used, double slope, double p, bool fast) {
  c10::OptionalDeviceGuard guard(x.device());
  auto gp = csr_plan(row_ptr, col_ind, col_ptr, row_ind, permute);
  return gat_backward(*gp, gp->colT, gp->posT, el.contiguous(), er.contiguous(), x.contiguous(), grad, out, rmax, rden,
                      rng_used, slope, p, fast);
}

// y = dropout(relu(a + bias))                                                      (gcn_conv.py:105-106, models/gcn.py:55-59)
static std::tuple<Tensor, Tensor> bias_act_forward_kernel(const Tensor &a_, const OptT_ &bias, bool relu, double p) {
  Tensor a = a_.contiguous(), b = opt(bias);
  same_device({&a, &b});
  f32("a", a);
  TORCH_CHECK(p >= 0.0 && p < 1.0, "p_drop must be in [0, 1)");
  c10::OptionalDeviceGuard guard(a.device());
  const Api &api = api_for(a.device());
  const int64_t N = a.dim() > 0 ? a.size(0) : 1, K = N > 0 ? a.numel() / N : width_of(a.sizes());
  if (b.defined()) {
    f32("bias", b);
    b = b.contiguous().reshape({-1});
    TORCH_CHECK(b.numel() == K, "bias must hold one value per column");
  }
  Tensor y = at::empty_like(a), rng, used = at::empty({0}, a.options().dtype(at::kLong));
  if (p > 0) {
    rng = rng_state(a.device());
    used = rng.clone();
  }
  check(api, api.ggl_bias_act_fwd(a.data_ptr<float>(), b.defined() ? b.data_ptr<float>() : nullptr, N, K, relu ? 1 : 0,
                                  static_cast<float>(p), rng.defined() ? rng.data_ptr<int64_t>() : nullptr,
                                  y.data_ptr<float>(), stream_of(a.device())));
  return {y, used};
}
// (ga, gbias): the mask is rebuilt from y, the bias gradient reduced in the same pass (epilogue.hip)
static std::tuple<Tensor, Tensor> bias_act_backward_kernel(const Tensor &g_, const Tensor &y, bool has_bias, bool relu, double p,
                                                           const Tensor &rng_used) {
  Tensor g = g_.contiguous();
  c10::OptionalDeviceGuard guard(g.device());
  const Api &api = api_for(g.device());
  const int64_t N = g.dim() > 0 ? g.size(0) : 1, K = N > 0 ? g.numel() / N : width_of(g.sizes());
  if (!relu && p <= 0 && !has_bias) return {g, Tensor()};
  Tensor ga = at::empty_like(g), gb = has_bias ? at::empty({K}, g.options()) : Tensor();
  const size_t wsb = api.ggl_bias_act_bwd_workspace_bytes(N, K);
  Tensor ws = at::empty({static_cast<int64_t>(std::max<size_t>(wsb, 4))}, g.options().dtype(at::kByte));
  check(api, api.ggl_bias_act_bwd(g.data_ptr<float>(), y.data_ptr<float>(), N, K, relu ? 1 : 0, static_cast<float>(p),
                                  (p > 0 && rng_used.numel() == 2) ? rng_used.data_ptr<int64_t>() : nullptr,
                                  ga.data_ptr<float>(), gb.defined() ? gb.data_ptr<float>() : nullptr, ws.data_ptr(), wsb,
                                  stream_of(g.device())));
  return {ga, gb.defined() ? gb : at::empty({0}, g.options())};
}

// y = dropout(relu(reduce_{j -> i} w x_j + add_i + bias)) in ONE kernel (reduce.hip MODE_SPMM_EPI): 2-D x, K % 4 == 0
static std::tuple<Tensor, Tensor> spmm_epi_forward_kernel(const Tensor &index, const OptT_ &weight, const Tensor &x, bool mean,
                                                          const OptT_ &add_, const OptT_ &bias_, bool relu, double p) {
  c10::OptionalDeviceGuard guard(x.device());
  SpArgs s = spmm_args(index, weight, x);
  Tensor add = opt(add_), bias = opt(bias_);
  same_device({&x, &add, &bias});
  TORCH_CHECK(s.x.dim() == 2 && s.x.size(1) % 4 == 0, "spmm_epi_forward needs a 2-D x whose width is a multiple of 4");
  TORCH_CHECK(p >= 0.0 && p < 1.0, "p_drop must be in [0, 1)");
  const Api &a = api_for(x.device());
  const int64_t K = s.x.size(1);
  if (add.defined()) {
    f32("add", add);
    TORCH_CHECK(add.dim() == 2 && add.size(0) == s.gp->N_dst && add.size(1) == K, "add must be [destination rows, feature width]");
    add = add.contiguous();
  }
  if (bias.defined()) {
    f32("bias", bias);
    bias = bias.contiguous().reshape({-1});
    TORCH_CHECK(bias.numel() == K, "bias must hold one value per column");
  }
  const SegPlan &p_ = *s.gp->fwd;
  Tensor y = at::empty({s.gp->N_dst, K}, s.x.options());
  Tensor part = partial_for(a, p_, s.x, K, false);
  ggl_segplan_t cs = p_.c(part);
  Tensor keep, rng, used = at::empty({0}, s.x.options().dtype(at::kLong));
  auto [wp, by_pos] = weights_for(a, *s.gp, p_, s.w, keep);
  if (p > 0) {
    rng = rng_state(x.device());
    used = rng.clone();
  }
  check(a, a.ggl_spmm_epi_ex(&cs, s.gp->col.data_ptr<int32_t>(), wp, by_pos, s.x.data_ptr<float>(), K, K, y.data_ptr<float>(),
                             K, 0, mean ? 1 : 0, add.defined() ? add.data_ptr<float>() : nullptr, add.defined() ? K : 0,
                             bias.defined() ? bias.data_ptr<float>() : nullptr, relu ? 1 : 0, static_cast<float>(p),
                             rng.defined() ? rng.data_ptr<int64_t>() : nullptr, 0, 0, 1, stream_of(x.device())));
  return {y, used};
}

// relu(segment_{sum,mean}(x, ids, N) + add + bias) for f32 messages [E, K] in one kernel (sage_conv.py:100-108)
static Tensor segment_epi_forward_kernel(const Tensor &x_, const Tensor &index, int64_t N, bool mean, const OptT_ &add_,
                                         const OptT_ &bias_, bool relu) {
  seg_args(x_, index);
  Tensor x = x_.contiguous(), add = opt(add_), bias = opt(bias_);
  same_device({&x, &add, &bias});
  f32("msg", x);
  TORCH_CHECK(x.dim() == 2, "segment_epi expects [E, K] messages");
  c10::OptionalDeviceGuard guard(x.device());
  const Api &a = api_for(x.device());
  auto plan = seg_plan(index, N);
  const int64_t K = x.size(1);
  if (add.defined()) {
    f32("add", add);
    TORCH_CHECK(add.dim() == 2 && add.size(0) == N && add.size(1) == K, "add must be [num_segments, feature width]");
    add = add.contiguous();
  }
  if (bias.defined()) {
    f32("bias", bias);
    bias = bias.contiguous().reshape({-1});
    TORCH_CHECK(bias.numel() == K, "bias must hold one value per column");
  }
  Tensor y = at::empty({N, K}, x.options());
  Tensor part = partial_for(a, *plan, x, K, false);
  ggl_segplan_t cs = plan->c(part);
  check(a, a.ggl_segment_epi(x.data_ptr<float>(), &cs, K, mean ? 1 : 0, add.defined() ? add.data_ptr<float>() : nullptr, 0,
                             bias.defined() ? bias.data_ptr<float>() : nullptr, relu ? 1 : 0, 0.0f, nullptr,
                             y.data_ptr<float>(), stream_of(x.device())));
  return y;
}

// One sampled hop with fixed capacities and device-side sizes (sample.hip ggl_sample_hop; ops/sparse/cpu/sample.cpp:10-135):
// nothing is read back, so a mini-batch step captures into one hipGraph.  Returns (rowptr [B_cap + 1], col [E_cap] int32
// local ids, e_id [E_cap], n_id [S_cap], counts [3] = {nodes, edges, overflow}); first_pos is the caller's relabel scratch
// (one int64 per graph node, all 2^62 on entry and on exit).
static std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> sample_hop_kernel(const Tensor &rowptr, const Tensor &col,
                                                                            const Tensor &seeds, const Tensor &n_seeds,
                                                                            int64_t num_nodes, int64_t fanout, int64_t e_cap,
                                                                            int64_t s_cap, Tensor &first_pos) {
  same_device({&rowptr, &col, &seeds, &n_seeds, &first_pos});
  const Tensor *all5[5] = {&rowptr, &col, &seeds, &n_seeds, &first_pos};
  for (const Tensor *t : all5)
    TORCH_CHECK(t->scalar_type() == at::kLong && t->is_contiguous(), "sample_hop takes contiguous int64 tensors");
  TORCH_CHECK(fanout > 0 && first_pos.numel() >= num_nodes && rowptr.numel() == num_nodes + 1, "bad sample_hop arguments");
  c10::OptionalDeviceGuard guard(rowptr.device());
  const Api &a = api_for(rowptr.device());
  const int64_t b_cap = seeds.size(0);
  auto i64 = rowptr.options();
  Tensor out_rowptr = at::empty({b_cap + 1}, i64), out_col = at::empty({std::max<int64_t>(e_cap, 1)}, i64.dtype(at::kInt));
  Tensor out_eid = at::empty({std::max<int64_t>(e_cap, 1)}, i64), out_nid = at::empty({s_cap}, i64), counts = at::empty({3}, i64);
  const size_t wsb = a.ggl_sample_hop_workspace_bytes(b_cap, e_cap);
  Tensor ws = at::empty({static_cast<int64_t>(wsb)}, i64.dtype(at::kByte));
  Tensor rng = rng_state(rowptr.device());
  check(a, a.ggl_sample_hop(rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(), seeds.data_ptr<int64_t>(),
                            n_seeds.data_ptr<int64_t>(), b_cap, num_nodes, fanout, e_cap, s_cap, rng.data_ptr<int64_t>(),
                            first_pos.data_ptr<int64_t>(), out_rowptr.data_ptr<int64_t>(), out_col.data_ptr<int32_t>(),
                            out_eid.data_ptr<int64_t>(), out_nid.data_ptr<int64_t>(), counts.data_ptr<int64_t>(), ws.data_ptr(),
                            wsb, stream_of(rowptr.device())));
  return {out_rowptr, out_col, out_eid, out_nid, counts};
}

// ---------------------------------------------------------------------------------------------------------------
// Autograd: every formula re-dispatches (below the autograd key) to the ops above
// ---------------------------------------------------------------------------------------------------------------
template <typename Sig>
static auto op_handle(const char *name) {
  return c10::Dispatcher::singleton().findSchemaOrThrow(name, "").typed<Sig>();
}
using OptT = c10::optional<Tensor>;
using SegSig = Tensor(const Tensor &, const Tensor &, int64_t);
using SpSig = Tensor(const Tensor &, const OptT &, const Tensor &);

template <Red OP>
struct SegmentFn : public torch::autograd::Function<SegmentFn<OP>> {
  static variable_list forward(AutogradContext *ctx, const Tensor &x, const Tensor &index, int64_t N) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->saved_data["x_shape"] = x.sizes().vec();
    ctx->saved_data["N"] = N;
    if (OP == Red::Max) {
      static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, int64_t)>("ggl::segment_max");
      auto r = op.call(x, index, N);
      ctx->save_for_backward({std::get<1>(r)});
      ctx->mark_non_differentiable({std::get<1>(r)});
      return {std::get<0>(r), std::get<1>(r)};
    }
    static auto op = op_handle<SegSig>(OP == Red::Sum ? "ggl::segment_sum" : "ggl::segment_mean");
    ctx->save_for_backward({index});
    return {op.call(x, index, N)};
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    auto shape = ctx->saved_data["x_shape"].toIntVector();
    Tensor gin;
    if (OP == Red::Sum) {
      static auto op = op_handle<Tensor(const Tensor &, const Tensor &, c10::IntArrayRef)>("ggl::segment_sum_backward");
      gin = op.call(grads[0], saved[0], shape);
    } else if (OP == Red::Mean) {
      static auto op = op_handle<Tensor(const Tensor &, const Tensor &, int64_t, c10::IntArrayRef)>("ggl::segment_mean_backward");
      gin = op.call(grads[0], saved[0], ctx->saved_data["N"].toInt(), shape);
    } else {
      static auto op = op_handle<Tensor(const Tensor &, const Tensor &, c10::IntArrayRef)>("ggl::segment_max_backward");
      gin = op.call(grads[0], saved[0], shape);
    }
    return {gin, Tensor(), Tensor()};
  }
};

template <SpOp OP>
struct SpMMFn : public torch::autograd::Function<SpMMFn<OP>> {
  static Tensor forward(AutogradContext *ctx, const Tensor &index, const OptT &weight, const Tensor &x) {
    at::AutoDispatchBelowADInplaceOrView below;
    // index and weight are not differentiable (gspmm.cpp:30); the arg-max of `max` is a source NODE id
    if (OP == SpOp::Max) {
      static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor &, const OptT &, const Tensor &)>("ggl::spmm_max_arg");
      auto r = op.call(index, weight, x);
      ctx->save_for_backward({index, opt(weight), std::get<1>(r)});
      return std::get<0>(r);
    }
    static auto op = op_handle<SpSig>(OP == SpOp::Sum ? "ggl::spmm_sum" : "ggl::spmm_mean");
    ctx->save_for_backward({index, opt(weight)});
    return op.call(index, weight, x);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    OptT w = saved[1].defined() ? OptT(saved[1]) : OptT();
    Tensor gx;
    if (OP == SpOp::Max) {
      static auto op = op_handle<Tensor(const Tensor &, const OptT &, const Tensor &, const Tensor &)>("ggl::spmm_max_backward");
      gx = op.call(saved[0], w, grads[0], saved[2]);
    } else {
      static auto op = op_handle<SpSig>(OP == SpOp::Sum ? "ggl::spmm_sum_backward" : "ggl::spmm_mean_backward");
      gx = op.call(saved[0], w, grads[0]);
    }
    return {Tensor(), Tensor(), gx};
  }
};

struct BSpMMFn : public torch::autograd::Function<BSpMMFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &index, const Tensor &weight, const Tensor &x) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = op_handle<Tensor(const Tensor &, const Tensor &, const Tensor &)>("ggl::bspmm_sum");
    ctx->save_for_backward({index, weight, x});
    return op.call(index, weight, x);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &)>(
        "ggl::bspmm_sum_backward");
    auto r = op.call(saved[0], saved[1], saved[2], grads[0]);
    // the reference returns grad_weight although it marked weight non-differentiable (gspmm.cpp:208,259)
    return {Tensor(), std::get<0>(r), std::get<1>(r)};
  }
};

static Tensor segment_sum_autograd(const Tensor &x, const Tensor &i, int64_t N) { return SegmentFn<Red::Sum>::apply(x, i, N)[0]; }
static Tensor segment_mean_autograd(const Tensor &x, const Tensor &i, int64_t N) { return SegmentFn<Red::Mean>::apply(x, i, N)[0]; }
static std::tuple<Tensor, Tensor> segment_max_autograd(const Tensor &x, const Tensor &i, int64_t N) {
  auto r = SegmentFn<Red::Max>::apply(x, i, N);
  return {r[0], r[1]};
}
static Tensor spmm_sum_autograd(const Tensor &i, const OptT &w, const Tensor &x) { return SpMMFn<SpOp::Sum>::apply(i, w, x); }
static Tensor spmm_mean_autograd(const Tensor &i, const OptT &w, const Tensor &x) { return SpMMFn<SpOp::Mean>::apply(i, w, x); }
static Tensor spmm_max_autograd(const Tensor &i, const OptT &w, const Tensor &x) { return SpMMFn<SpOp::Max>::apply(i, w, x); }
static Tensor bspmm_sum_autograd(const Tensor &i, const Tensor &w, const Tensor &x) { return BSpMMFn::apply(i, w, x); }

// ---- autograd of the fused route --------------------------------------------------------------------------------
using GatFwdSig = std::tuple<Tensor, Tensor, Tensor, Tensor, bool>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                                    double, int64_t, double);
using GatBwdSig = std::tuple<Tensor, Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                     const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                     const Tensor &, double, int64_t, double, bool);
struct GatFn : public torch::autograd::Function<GatFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &index, const Tensor &el, const Tensor &er, const Tensor &x,
                        double slope, int64_t n, double p) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = op_handle<GatFwdSig>("ggl::gat_fused_forward");
    auto r = op.call(index, el, er, x, slope, n, p);
    ctx->save_for_backward({index, el, er, x, std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r)});
    ctx->saved_data["slope"] = slope;
    ctx->saved_data["n"] = n;
    ctx->saved_data["p"] = p;
    ctx->saved_data["fast"] = std::get<4>(r);
    return std::get<0>(r);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    static auto op = op_handle<GatBwdSig>("ggl::gat_fused_backward");
    auto r = op.call(s[0], s[1], s[2], s[3], grads[0], s[4], s[5], s[6], s[7], ctx->saved_data["slope"].toDouble(),
                     ctx->saved_data["n"].toInt(), ctx->saved_data["p"].toDouble(), ctx->saved_data["fast"].toBool());
    return {Tensor(), std::get<0>(r), std::get<1>(r), std::get<2>(r), Tensor(), Tensor(), Tensor()};
  }
};
static Tensor gat_fused_autograd(const Tensor &index, const Tensor &el, const Tensor &er, const Tensor &x, double slope,
                                 c10::optional<int64_t> num_nodes, double p) {
  gat_check(el, er, x);
  const int64_t n = num_nodes.has_value() ? *num_nodes : x.size(0), C = x.size(2);
  // 41 classes per head: one zero-padded copy keeps every walk on 16-byte slices (ops.py Engine.gat_fused); the pad
  // channels aggregate to zero and are dropped — pad and slice are ordinary differentiable ops
  const int64_t Cp = index.dim() == 2 ? api_for(x.device()).ggl_policy_head_channels(C, index.size(1), x.size(0)) : C;
  if (Cp != C) {
    Tensor xp = at::constant_pad_nd(x, {0, Cp - C});
    return GatFn::apply(index, el, er, xp, slope, n, p).slice(2, 0, C);
  }
  return GatFn::apply(index, el, er, x, slope, n, p);
}

using GatCsrFwdSig = std::tuple<Tensor, Tensor, Tensor, Tensor, bool>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                                       const Tensor &, const Tensor &, const Tensor &,
                                                                       const Tensor &, double, double);
using GatCsrBwdSig = std::tuple<Tensor, Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                        const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                        const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                        const Tensor &, double, double, bool);
struct GatCsrFn : public torch::autograd::Function<GatCsrFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &rp, const Tensor &ci, const Tensor &cp, const Tensor &ri,
                        const Tensor &pm, const Tensor &el, const Tensor &er, const Tensor &x, double slope, double p) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = op_handle<GatCsrFwdSig>("ggl::gat_fused_csr_forward");
    auto r = op.call(rp, ci, cp, ri, pm, el, er, x, slope, p);
    ctx->save_for_backward({rp, ci, cp, ri, pm, el, er, x, std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r)});
    ctx->saved_data["slope"] = slope;
    ctx->saved_data["p"] = p;
    ctx->saved_data["fast"] = std::get<4>(r);
    return std::get<0>(r);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    static auto op = op_handle<GatCsrBwdSig>("ggl::gat_fused_csr_backward");
    auto r = op.call(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], grads[0], s[8], s[9], s[10], s[11],
                     ctx->saved_data["slope"].toDouble(), ctx->saved_data["p"].toDouble(), ctx->saved_data["fast"].toBool());
    return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), std::get<0>(r), std::get<1>(r), std::get<2>(r), Tensor(), Tensor()};
  }
};
static Tensor gat_fused_csr_autograd(const Tensor &rp, const Tensor &ci, const Tensor &cp, const Tensor &ri, const Tensor &pm,
                                     const Tensor &el, const Tensor &er, const Tensor &x, double slope, double p) {
  gat_check(el, er, x);
  const int64_t C = x.size(2);
  const int64_t Cp = api_for(x.device()).ggl_policy_head_channels(C, ci.size(0), x.size(0));
  if (Cp != C) {
    Tensor xp = at::constant_pad_nd(x, {0, Cp - C});
    return GatCsrFn::apply(rp, ci, cp, ri, pm, el, er, xp, slope, p).slice(2, 0, C);
  }
  return GatCsrFn::apply(rp, ci, cp, ri, pm, el, er, x, slope, p);
}

using BiasFwdSig = std::tuple<Tensor, Tensor>(const Tensor &, const OptT_ &, bool, double);
using BiasBwdSig = std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, bool, bool, double, const Tensor &);
struct BiasActFn : public torch::autograd::Function<BiasActFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &a, const OptT_ &bias, bool relu, double p) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = op_handle<BiasFwdSig>("ggl::bias_act_forward");
    auto r = op.call(a, bias, relu, p);
    ctx->save_for_backward({std::get<0>(r), std::get<1>(r)});
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["p"] = p;
    ctx->saved_data["bias_shape"] = opt(bias).defined() ? opt(bias).sizes().vec() : std::vector<int64_t>{};
    ctx->saved_data["has_bias"] = opt(bias).defined();
    return std::get<0>(r);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    static auto op = op_handle<BiasBwdSig>("ggl::bias_act_backward");
    const bool hb = ctx->saved_data["has_bias"].toBool();
    auto r = op.call(grads[0], s[0], hb, ctx->saved_data["relu"].toBool(), ctx->saved_data["p"].toDouble(), s[1]);
    Tensor gb = hb ? std::get<1>(r).reshape(ctx->saved_data["bias_shape"].toIntVector()) : Tensor();
    return {std::get<0>(r), gb, Tensor(), Tensor()};
  }
};
static Tensor bias_act_autograd(const Tensor &a, const OptT_ &bias, bool relu, double p) { return BiasActFn::apply(a, bias, relu, p); }

using EpiFwdSig = std::tuple<Tensor, Tensor>(const Tensor &, const OptT_ &, const Tensor &, bool, const OptT_ &, const OptT_ &, bool, double);
struct SpMMEpiFn : public torch::autograd::Function<SpMMEpiFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &index, const OptT_ &weight, const Tensor &x, bool mean,
                        const OptT_ &add, const OptT_ &bias, bool relu, double p) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = op_handle<EpiFwdSig>("ggl::spmm_epi_forward");
    auto r = op.call(index, weight, x, mean, add, bias, relu, p);
    ctx->save_for_backward({index, opt(weight), std::get<0>(r), std::get<1>(r)});
    ctx->saved_data["mean"] = mean;
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["p"] = p;
    ctx->saved_data["has_add"] = opt(add).defined();
    ctx->saved_data["has_bias"] = opt(bias).defined();
    ctx->saved_data["bias_shape"] = opt(bias).defined() ? opt(bias).sizes().vec() : std::vector<int64_t>{};
    return std::get<0>(r);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    static auto bop = op_handle<BiasBwdSig>("ggl::bias_act_backward");
    const bool hb = ctx->saved_data["has_bias"].toBool();
    auto r = bop.call(grads[0], s[2], hb, ctx->saved_data["relu"].toBool(), ctx->saved_data["p"].toDouble(), s[3]);
    Tensor ga = std::get<0>(r);
    OptT_ w = s[1].defined() ? OptT_(s[1]) : OptT_();
    static auto sum_bwd = op_handle<SpSig>("ggl::spmm_sum_backward");
    static auto mean_bwd = op_handle<SpSig>("ggl::spmm_mean_backward");
    Tensor gx = ctx->saved_data["mean"].toBool() ? mean_bwd.call(s[0], w, ga) : sum_bwd.call(s[0], w, ga);
    Tensor gb = hb ? std::get<1>(r).reshape(ctx->saved_data["bias_shape"].toIntVector()) : Tensor();
    return {Tensor(), Tensor(), gx, Tensor(), ctx->saved_data["has_add"].toBool() ? ga : Tensor(), gb, Tensor(), Tensor()};
  }
};
// one kernel for 16-byte rows; the reduce op followed by the adds and the epilogue pass otherwise (same values)
static Tensor spmm_epi_autograd(const Tensor &index, const OptT_ &weight, const Tensor &x, bool mean, const OptT_ &add,
                                const OptT_ &bias, bool relu, double p) {
  if (x.dim() == 2 && x.size(1) % 4 == 0) return SpMMEpiFn::apply(index, weight, x, mean, add, bias, relu, p);
  Tensor out = mean ? spmm_mean_autograd(index, weight, x) : spmm_sum_autograd(index, weight, x);
  if (opt(add).defined()) out = out + *add;
  return BiasActFn::apply(out, bias, relu, p);
}

struct SegEpiFn : public torch::autograd::Function<SegEpiFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &x, const Tensor &index, int64_t N, bool mean, const OptT_ &add,
                        const OptT_ &bias, bool relu) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = op_handle<Tensor(const Tensor &, const Tensor &, int64_t, bool, const OptT_ &, const OptT_ &, bool)>(
        "ggl::segment_epi_forward");
    Tensor y = op.call(x, index, N, mean, add, bias, relu);
    ctx->save_for_backward({index, y});
    ctx->saved_data["x_shape"] = x.sizes().vec();
    ctx->saved_data["N"] = N;
    ctx->saved_data["mean"] = mean;
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_add"] = opt(add).defined();
    ctx->saved_data["has_bias"] = opt(bias).defined();
    ctx->saved_data["bias_shape"] = opt(bias).defined() ? opt(bias).sizes().vec() : std::vector<int64_t>{};
    return y;
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    static auto bop = op_handle<BiasBwdSig>("ggl::bias_act_backward");
    const bool hb = ctx->saved_data["has_bias"].toBool();
    auto r = bop.call(grads[0], s[1], hb, ctx->saved_data["relu"].toBool(), 0.0, at::empty({0}, s[0].options()));
    Tensor ga = std::get<0>(r);
    auto shape = ctx->saved_data["x_shape"].toIntVector();
    Tensor gx;
    if (ctx->saved_data["mean"].toBool()) {
      static auto op = op_handle<Tensor(const Tensor &, const Tensor &, int64_t, c10::IntArrayRef)>("ggl::segment_mean_backward");
      gx = op.call(ga, s[0], ctx->saved_data["N"].toInt(), shape);
    } else {
      static auto op = op_handle<Tensor(const Tensor &, const Tensor &, c10::IntArrayRef)>("ggl::segment_sum_backward");
      gx = op.call(ga, s[0], shape);
    }
    Tensor gb = hb ? std::get<1>(r).reshape(ctx->saved_data["bias_shape"].toIntVector()) : Tensor();
    return {gx, Tensor(), Tensor(), Tensor(), ctx->saved_data["has_add"].toBool() ? ga : Tensor(), gb, Tensor()};
  }
};
static Tensor segment_epi_autograd(const Tensor &x, const Tensor &index, int64_t N, bool mean, const OptT_ &add,
                                   const OptT_ &bias, bool relu) {
  return SegEpiFn::apply(x, index, N, mean, add, bias, relu);
}

// ---------------------------------------------------------------------------------------------------------------
// Meta (shapes / dtypes only) and housekeeping ops
// ---------------------------------------------------------------------------------------------------------------
static Tensor seg_meta(const Tensor &x, const Tensor &, int64_t N) { return at::empty(out_shape(x, N), x.options()); }
static std::tuple<Tensor, Tensor> seg_max_meta(const Tensor &x, const Tensor &, int64_t N) {
  return {at::empty(out_shape(x, N), x.options()), at::empty(out_shape(x, N), x.options().dtype(at::kLong))};
}
static Tensor like_x_meta(const Tensor &, const OptT &, const Tensor &x) { return at::empty_like(x); }
static std::tuple<Tensor, Tensor> spmm_max_arg_meta(const Tensor &, const OptT &, const Tensor &x) {
  return {at::empty_like(x), at::empty(x.sizes(), x.options().dtype(at::kLong))};
}
static Tensor bspmm_meta(const Tensor &, const Tensor &, const Tensor &x) { return at::empty_like(x); }
static Tensor seg_bwd_meta(const Tensor &g, const Tensor &, c10::IntArrayRef shape) { return at::empty(shape, g.options()); }
static Tensor seg_mean_bwd_meta(const Tensor &g, const Tensor &, int64_t, c10::IntArrayRef shape) {
  return at::empty(shape, g.options());
}
static Tensor spmm_max_bwd_meta(const Tensor &, const OptT &, const Tensor &g, const Tensor &) { return at::empty_like(g); }
static std::tuple<Tensor, Tensor> bspmm_bwd_meta(const Tensor &, const Tensor &w, const Tensor &x, const Tensor &) {
  return {at::empty_like(w), at::empty_like(x)};
}

static std::tuple<Tensor, Tensor, Tensor, Tensor, bool> gat_fwd_meta(const Tensor &, const Tensor &, const Tensor &er,
                                                                     const Tensor &x, double, int64_t n, double) {
  auto o = x.options();
  return {at::empty({n, x.size(1), x.size(2)}, o), at::empty({n, x.size(1)}, o), at::empty({n, x.size(1)}, o),
          at::empty({0}, o.dtype(at::kLong)), false};
}
static std::tuple<Tensor, Tensor, Tensor> gat_bwd_meta(const Tensor &, const Tensor &el, const Tensor &er, const Tensor &x,
                                                       const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                       const Tensor &, double, int64_t, double, bool) {
  return {at::empty_like(el), at::empty_like(er), at::empty_like(x)};
}
static Tensor gat_meta(const Tensor &, const Tensor &, const Tensor &, const Tensor &x, double, c10::optional<int64_t> n, double) {
  return at::empty({n.has_value() ? *n : x.size(0), x.size(1), x.size(2)}, x.options());
}
static std::tuple<Tensor, Tensor, Tensor, Tensor, bool> gat_csr_fwd_meta(const Tensor &rp, const Tensor &, const Tensor &,
                                                                         const Tensor &, const Tensor &, const Tensor &,
                                                                         const Tensor &, const Tensor &x, double, double) {
  const int64_t n = rp.size(0) - 1;
  auto o = x.options();
  return {at::empty({n, x.size(1), x.size(2)}, o), at::empty({n, x.size(1)}, o), at::empty({n, x.size(1)}, o),
          at::empty({0}, o.dtype(at::kLong)), false};
}
static std::tuple<Tensor, Tensor, Tensor> gat_csr_bwd_meta(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                           const Tensor &, const Tensor &el, const Tensor &er, const Tensor &x,
                                                           const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                           const Tensor &, double, double, bool) {
  return {at::empty_like(el), at::empty_like(er), at::empty_like(x)};
}
static Tensor gat_csr_meta(const Tensor &rp, const Tensor &, const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                           const Tensor &, const Tensor &x, double, double) {
  return at::empty({rp.size(0) - 1, x.size(1), x.size(2)}, x.options());
}
static std::tuple<Tensor, Tensor> bias_fwd_meta(const Tensor &a, const OptT_ &, bool, double) {
  return {at::empty_like(a), at::empty({0}, a.options().dtype(at::kLong))};
}
static std::tuple<Tensor, Tensor> bias_bwd_meta(const Tensor &g, const Tensor &, bool hb, bool, double, const Tensor &) {
  const int64_t N = g.dim() > 0 ? g.size(0) : 1;
  return {at::empty_like(g), at::empty({hb && N > 0 ? g.numel() / N : 0}, g.options())};
}
static Tensor bias_meta(const Tensor &a, const OptT_ &, bool, double) { return at::empty_like(a); }
static std::tuple<Tensor, Tensor> epi_fwd_meta(const Tensor &, const OptT_ &, const Tensor &x, bool, const OptT_ &, const OptT_ &,
                                               bool, double) {
  return {at::empty_like(x), at::empty({0}, x.options().dtype(at::kLong))};
}
static Tensor epi_meta(const Tensor &, const OptT_ &, const Tensor &x, bool, const OptT_ &, const OptT_ &, bool, double) {
  return at::empty_like(x);
}
static Tensor seg_epi_meta(const Tensor &x, const Tensor &, int64_t N, bool, const OptT_ &, const OptT_ &, bool) {
  return at::empty(out_shape(x, N), x.options());
}

static void clear_caches() {
  seg_cache().clear();
  graph_cache().clear();
  csr_cache().clear();
}
static std::vector<int64_t> plan_stats() {
  return {static_cast<int64_t>(g_plans_built.load()), static_cast<int64_t>(g_plan_hits.load())};
}

}  // namespace ggl_torch

TORCH_LIBRARY(ggl, m) {
  m.def("segment_sum(Tensor x, Tensor index, int N) -> Tensor");
  m.def("segment_mean(Tensor x, Tensor index, int N) -> Tensor");
  m.def("segment_max(Tensor x, Tensor index, int N) -> (Tensor, Tensor)");
  m.def("spmm_sum(Tensor index, Tensor? weight, Tensor x) -> Tensor");
  m.def("spmm_mean(Tensor index, Tensor? weight, Tensor x) -> Tensor");
  m.def("spmm_max(Tensor index, Tensor? weight, Tensor x) -> Tensor");
  m.def("bspmm_sum(Tensor index, Tensor weight, Tensor x) -> Tensor");
  // backward passes and the arg-returning max (what the autograd formulas call; usable on their own)
  m.def("segment_sum_backward(Tensor grad, Tensor index, int[] x_shape) -> Tensor");
  m.def("segment_mean_backward(Tensor grad, Tensor index, int N, int[] x_shape) -> Tensor");
  m.def("segment_max_backward(Tensor grad, Tensor arg, int[] x_shape) -> Tensor");
  m.def("spmm_sum_backward(Tensor index, Tensor? weight, Tensor grad) -> Tensor");
  m.def("spmm_mean_backward(Tensor index, Tensor? weight, Tensor grad) -> Tensor");
  m.def("spmm_max_arg(Tensor index, Tensor? weight, Tensor x) -> (Tensor, Tensor)");
  m.def("spmm_max_backward(Tensor index, Tensor? weight, Tensor grad, Tensor arg) -> Tensor");
  m.def("bspmm_sum_backward(Tensor index, Tensor weight, Tensor x, Tensor grad) -> (Tensor, Tensor)");
  // the fused route: edge-softmax + aggregate (COO edge list / the caller's CSR = dgNN's GATConvFuse), the layer epilogue
  // alone and inside the aggregate's store, one static-shape sampler hop
  m.def("gat_fused(Tensor index, Tensor el, Tensor er, Tensor x, float negative_slope=0.2, int? num_nodes=None, "
        "float dropout_rate=0.0) -> Tensor");
  m.def("gat_fused_forward(Tensor index, Tensor el, Tensor er, Tensor x, float negative_slope, int num_nodes, "
        "float dropout_rate) -> (Tensor, Tensor, Tensor, Tensor, bool)");
  m.def("gat_fused_backward(Tensor index, Tensor el, Tensor er, Tensor x, Tensor grad, Tensor out, Tensor rowmax, "
        "Tensor rowden, Tensor rng_used, float negative_slope, int num_nodes, float dropout_rate, bool fast) -> "
        "(Tensor, Tensor, Tensor)");
  m.def("gat_fused_csr(Tensor row_ptr, Tensor col_ind, Tensor col_ptr, Tensor row_ind, Tensor permute, Tensor el, "
        "Tensor er, Tensor x, float negative_slope=0.2, float dropout_rate=0.0) -> Tensor");
  m.def("gat_fused_csr_forward(Tensor row_ptr, Tensor col_ind, Tensor col_ptr, Tensor row_ind, Tensor permute, Tensor el, "
        "Tensor er, Tensor x, float negative_slope, float dropout_rate) -> (Tensor, Tensor, Tensor, Tensor, bool)");
  m.def("gat_fused_csr_backward(Tensor row_ptr, Tensor col_ind, Tensor col_ptr, Tensor row_ind, Tensor permute, Tensor el, "
        "Tensor er, Tensor x, Tensor grad, Tensor out, Tensor rowmax, Tensor rowden, Tensor rng_used, float negative_slope, "
        "float dropout_rate, bool fast) -> (Tensor, Tensor, Tensor)");
  m.def("bias_act(Tensor a, Tensor? bias, bool relu, float p_drop) -> Tensor");
  m.def("bias_act_forward(Tensor a, Tensor? bias, bool relu, float p_drop) -> (Tensor, Tensor)");
  m.def("bias_act_backward(Tensor grad, Tensor y, bool has_bias, bool relu, float p_drop, Tensor rng_used) -> (Tensor, Tensor)");
  m.def("spmm_epi(Tensor index, Tensor? weight, Tensor x, bool mean=False, Tensor? add=None, Tensor? bias=None, "
        "bool relu=False, float p_drop=0.0) -> Tensor");
  m.def("spmm_epi_forward(Tensor index, Tensor? weight, Tensor x, bool mean, Tensor? add, Tensor? bias, bool relu, "
        "float p_drop) -> (Tensor, Tensor)");
  m.def("segment_epi(Tensor x, Tensor index, int N, bool mean=True, Tensor? add=None, Tensor? bias=None, bool relu=False) -> Tensor");
  m.def("segment_epi_forward(Tensor x, Tensor index, int N, bool mean, Tensor? add, Tensor? bias, bool relu) -> Tensor");
  m.def("sample_hop(Tensor rowptr, Tensor col, Tensor seeds, Tensor n_seeds, int num_nodes, int fanout, int e_cap, int s_cap, "
        "Tensor(a!) first_pos) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("reseed() -> ()", ggl_torch::reseed);
  m.def("clear_caches() -> ()", ggl_torch::clear_caches);
  m.def("plan_stats() -> int[]", ggl_torch::plan_stats);
}

#define GGL_BACKEND(KEY)                                     \
  TORCH_LIBRARY_IMPL(ggl, KEY, m) {                          \
    m.impl("segment_sum", ggl_torch::segment_sum_kernel);    \
    m.impl("segment_mean", ggl_torch::segment_mean_kernel);  \
    m.impl("segment_max", ggl_torch::segment_max_kernel);    \
    m.impl("spmm_sum", ggl_torch::spmm_sum_kernel);          \
    m.impl("spmm_mean", ggl_torch::spmm_mean_kernel);        \
    m.impl("spmm_max", ggl_torch::spmm_max_kernel);          \
    m.impl("bspmm_sum", ggl_torch::bspmm_sum_kernel);        \
    m.impl("segment_sum_backward", ggl_torch::segment_sum_backward_kernel);    \
    m.impl("segment_mean_backward", ggl_torch::segment_mean_backward_kernel);  \
    m.impl("segment_max_backward", ggl_torch::segment_max_backward_kernel);    \
    m.impl("spmm_sum_backward", ggl_torch::spmm_sum_backward_kernel);          \
    m.impl("spmm_mean_backward", ggl_torch::spmm_mean_backward_kernel);        \
    m.impl("spmm_max_arg", ggl_torch::spmm_max_arg_kernel);                    \
    m.impl("spmm_max_backward", ggl_torch::spmm_max_backward_kernel);          \
    m.impl("bspmm_sum_backward", ggl_torch::bspmm_sum_backward_kernel);        \
    m.impl("gat_fused_forward", ggl_torch::gat_fused_forward_kernel);          \
    m.impl("gat_fused_backward", ggl_torch::gat_fused_backward_kernel);        \
    m.impl("gat_fused_csr_forward", ggl_torch::gat_csr_forward_kernel);        \
    m.impl("gat_fused_csr_backward", ggl_torch::gat_csr_backward_kernel);      \
    m.impl("bias_act_forward", ggl_torch::bias_act_forward_kernel);            \
    m.impl("bias_act_backward", ggl_torch::bias_act_backward_kernel);          \
    m.impl("spmm_epi_forward", ggl_torch::spmm_epi_forward_kernel);            \
    m.impl("segment_epi_forward", ggl_torch::segment_epi_forward_kernel);      \
    m.impl("sample_hop", ggl_torch::sample_hop_kernel);                        \
  }
GGL_BACKEND(CPU)
GGL_BACKEND(CUDA)

TORCH_LIBRARY_IMPL(ggl, Autograd, m) {
  m.impl("segment_sum", ggl_torch::segment_sum_autograd);
  m.impl("segment_mean", ggl_torch::segment_mean_autograd);
  m.impl("segment_max", ggl_torch::segment_max_autograd);
  m.impl("spmm_sum", ggl_torch::spmm_sum_autograd);
  m.impl("spmm_mean", ggl_torch::spmm_mean_autograd);
  m.impl("spmm_max", ggl_torch::spmm_max_autograd);
  m.impl("bspmm_sum", ggl_torch::bspmm_sum_autograd);
  m.impl("gat_fused", ggl_torch::gat_fused_autograd);
  m.impl("gat_fused_csr", ggl_torch::gat_fused_csr_autograd);
  m.impl("bias_act", ggl_torch::bias_act_autograd);
  m.impl("spmm_epi", ggl_torch::spmm_epi_autograd);
  m.impl("segment_epi", ggl_torch::segment_epi_autograd);
}

TORCH_LIBRARY_IMPL(ggl, Meta, m) {
  m.impl("segment_sum", ggl_torch::seg_meta);
  m.impl("segment_mean", ggl_torch::seg_meta);
  m.impl("segment_max", ggl_torch::seg_max_meta);
  m.impl("spmm_sum", ggl_torch::like_x_meta);
  m.impl("spmm_mean", ggl_torch::like_x_meta);
  m.impl("spmm_max", ggl_torch::like_x_meta);
  m.impl("bspmm_sum", ggl_torch::bspmm_meta);
  m.impl("segment_sum_backward", ggl_torch::seg_bwd_meta);
  m.impl("segment_mean_backward", ggl_torch::seg_mean_bwd_meta);
  m.impl("segment_max_backward", ggl_torch::seg_bwd_meta);
  m.impl("spmm_sum_backward", ggl_torch::like_x_meta);
  m.impl("spmm_mean_backward", ggl_torch::like_x_meta);
  m.impl("spmm_max_arg", ggl_torch::spmm_max_arg_meta);
  m.impl("spmm_max_backward", ggl_torch::spmm_max_bwd_meta);
  m.impl("bspmm_sum_backward", ggl_torch::bspmm_bwd_meta);
  m.impl("gat_fused", ggl_torch::gat_meta);
  m.impl("gat_fused_forward", ggl_torch::gat_fwd_meta);
  m.impl("gat_fused_backward", ggl_torch::gat_bwd_meta);
  m.impl("gat_fused_csr", ggl_torch::gat_csr_meta);
  m.impl("gat_fused_csr_forward", ggl_torch::gat_csr_fwd_meta);
  m.impl("gat_fused_csr_backward", ggl_torch::gat_csr_bwd_meta);
  m.impl("bias_act", ggl_torch::bias_meta);
  m.impl("bias_act_forward", ggl_torch::bias_fwd_meta);
  m.impl("bias_act_backward", ggl_torch::bias_bwd_meta);
  m.impl("spmm_epi", ggl_torch::epi_meta);
  m.impl("spmm_epi_forward", ggl_torch::epi_fwd_meta);
  m.impl("segment_epi", ggl_torch::seg_epi_meta);
  m.impl("segment_epi_forward", ggl_torch::seg_epi_meta);
}
