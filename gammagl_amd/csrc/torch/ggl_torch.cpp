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
// What lives here besides the registrations is the HOST POLICY of the op library, the same one gammagl_amd/ops.py
// applies (the two are checked against each other bit for bit, tests/test_torch_cpp.py):
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
  X(ggl_bspmm_grad_w_sorted_scratch_bytes) X(ggl_bspmm_grad_w_sorted)

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
constexpr int64_t kMaxChunk = 4096, kMinChunk = 256, kResidentWaves = 256 * 32;   // MI355X: 256 CUs x 32 wavefronts

static int64_t auto_chunk(int64_t E) {   // the largest power of two <= E / resident waves, in [256, 4096]
  static const int64_t forced = env_int("GGL_LONG_ROW", 0);
  if (forced > 0) return forced;
  int64_t c = kMaxChunk;
  while (c > kMinChunk && c * kResidentWaves > E) c >>= 1;
  return c;
}

static Tensor row_order_of(const Tensor &counts);

static bool capturing(const c10::Device &d) {
  if (!d.is_cuda()) return false;
  c10::OptionalDeviceGuard guard(d);
  return c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None;
}

struct SegPlan {
  int64_t N = 0, E = 0, chunk = 0, n_long = 0, n_chunks = 0, max_len = 0, xcd_run = 0;
  bool sorted = false;
  uint64_t uid = 0;
  Tensor rowptr, perm, long_rows, chunk_ptr;
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
    s.xcd_run_rows = xcd_run;
    return s;
  }
  Tensor counts() const { return rowptr.slice(0, 1) - rowptr.slice(0, 0, N); }
};

static std::atomic<uint64_t> g_plans_built{0}, g_plan_hits{0};

// rows by descending length inside windows of consecutive ids, the heavy ones first (ops.py Engine._row_order)
static Tensor row_order_of(const Tensor &counts) {
  static const int64_t W = env_int("GGL_ROW_ORDER_WINDOW", 2048);
  const int64_t N = counts.size(0);
  if (W <= 0 || N <= W) return at::argsort(counts, /*stable=*/true, 0, /*descending=*/true).to(at::kInt);
  Tensor ar = at::arange(N, counts.options());
  Tensor group = at::where(counts >= 1024, at::zeros_like(ar), at::floor_divide(ar, W) + 1);
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
}

static std::shared_ptr<SegPlan> build_plan(const Tensor &ids_in, int64_t N) {
  const auto dev = ids_in.device();
  const Api &a = api_for(dev);
  Tensor ids = ids_in.contiguous();
  void *st = stream_of(dev);
  auto p = std::make_shared<SegPlan>();
  p->N = N;
  p->E = ids.size(0);
  p->chunk = auto_chunk(p->E);
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
    static const int64_t knob = env_int("GGL_XCD_RUN_ROWS", -1);
    int64_t run = knob;
    if (run < 0) run = (E >= (int64_t(1) << 22) && locality() > 0.5) ? 2048 : 0;
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
                                          const Tensor &x, int64_t n_out, const Tensor &aux = Tensor()) {
  const auto dev = x.device();
  const Api &a = api_for(dev);
  int64_t K = 1;
  for (int64_t d = 1; d < x.dim(); ++d) K *= x.size(d);
  const bool summing = op == SpOp::Sum || op == SpOp::Mean;
  if (summing && x.dim() == 2 && p.E >= 8 * x.size(0)) {
    int64_t pad = 0;
    if (K > 256 && K % 64 != 0) pad = (64 - K % 64) % 64;      // whole cache lines per 64-column block
    else if (K % 4 != 0 && K >= 8) pad = (4 - K % 4) % 4;      // 16-byte rows for the float4 kernels
    if (pad > 0) {
      Tensor xp = at::constant_pad_nd(x, {0, pad});
      auto r = spmm_fwd(op, gp, p, col, w, xp, n_out, aux);
      return {r.first.slice(1, 0, K).contiguous(), Tensor()};
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
      if (x.dim() == 2 && p.E >= 4 * x.size(0)) {
        // the division depends on the destination row only: once per row, then the plain transposed SpMM-sum
        Tensor cnt = (aux.slice(0, 1) - aux.slice(0, 0, aux.size(0) - 1)).clamp_min(1).to(at::kFloat).unsqueeze(1);
        Tensor xs = x / cnt;
        check(a, a.ggl_spmm_sum(&cs, c, wp, by_pos, xs.data_ptr<float>(), K, op_, st));
      } else {
        check(a, a.ggl_spmm_mean_bwd(&cs, c, wp, by_pos, xp, aux.data_ptr<int64_t>(), K, op_, st));
      }
      break;
    case SpOp::MaxBwd:
      check(a, a.ggl_spmm_max_bwd(&cs, c, wp, by_pos, xp, aux.data_ptr<int64_t>(), K, op_, st));
      break;
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

static bool bspmm_pads(const GraphPlan &gp, const Tensor &x) {
  const int64_t C = x.size(2);
  return C % 4 != 0 && C >= 8 && gp.E >= 8 * x.size(0);
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
  Tensor xin = bspmm_pads(*s.gp, s.x) ? at::constant_pad_nd(s.x, {0, (4 - C % 4) % 4}) : s.x;
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
  return spmm_fwd(SpOp::MaxBwd, *gp, *gp->bwd, gp->colT, w, g, gp->N_src, arg.contiguous()).first;
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
  const int64_t pad = bspmm_pads(*gp, x_in) ? (4 - C0 % 4) % 4 : 0;
  Tensor x = (pad > 0 ? at::constant_pad_nd(x_in, {0, pad}) : x_in).contiguous();
  Tensor g = (pad > 0 ? at::constant_pad_nd(grad, {0, pad}) : grad).contiguous();
  const int64_t C = C0 + pad;
  Tensor gx = bspmm_fwd(*gp, *gp->bwd, gp->colT, w, g, gp->N_src);
  Tensor gw = at::empty_like(w);
  void *st = stream_of(g.device());
  if (C % 4 == 0 && C > 16) {   // along the destination-sorted plan, strips staged through LDS (edgedot.hip)
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

static void clear_caches() {
  seg_cache().clear();
  graph_cache().clear();
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
}
