// The C++-registered operators called from a C++ program: no Python interpreter in the process.  Loads
// libggl_torch.so (argv[1]) with dlopen — its static initialisers register TORCH_LIBRARY(ggl) — looks the operators up
// in the dispatcher and checks spmm_sum / segment_max / their autograd against plain ATen on CPU tensors (the CPU key:
// libggl_mpops_host.so, resolved by the library from its own directory).  Prints "ok" and exits 0.
#include <ATen/ATen.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <dlfcn.h>

#include <cstdio>

int main(int argc, char **argv) {
  if (argc < 2 || dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL) == nullptr) {
    std::fprintf(stderr, "cannot load %s: %s\n", argc > 1 ? argv[1] : "(no path)", dlerror());
    return 2;
  }
  auto spmm = c10::Dispatcher::singleton().findSchemaOrThrow("ggl::spmm_sum", "")
                  .typed<at::Tensor(const at::Tensor &, const c10::optional<at::Tensor> &, const at::Tensor &)>();
  auto segmax = c10::Dispatcher::singleton().findSchemaOrThrow("ggl::segment_max", "")
                    .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor &, const at::Tensor &, int64_t)>();
  const int64_t N = 50, E = 600, K = 12;
  at::manual_seed(3);
  at::Tensor ei = at::randint(0, N, {2, E}, at::kLong);
  at::Tensor w = at::rand({E});
  at::Tensor x = at::randn({N, K}).set_requires_grad(true);
  at::Tensor y = spmm.call(ei, w, x);
  // the same sum with ATen: out[dst] += w * x[src]
  at::Tensor xr = x.detach().clone().set_requires_grad(true);
  at::Tensor ref = at::zeros({N, K}).index_add(0, ei[1], xr.index_select(0, ei[0]) * w.unsqueeze(1));
  if (!at::allclose(y, ref, 1e-5, 1e-5)) { std::fprintf(stderr, "spmm_sum forward differs\n"); return 1; }
  at::Tensor go = at::randn({N, K});
  y.backward(go);
  ref.backward(go);
  if (!at::allclose(x.grad(), xr.grad(), 1e-5, 1e-5)) { std::fprintf(stderr, "spmm_sum backward differs\n"); return 1; }
  at::Tensor msg = at::randn({E, 5});
  auto mx = segmax.call(msg, ei[1], N);
  at::Tensor want = at::full({N, 5}, -3.4028234663852886e38).scatter_reduce(0, ei[1].unsqueeze(1).expand({E, 5}), msg, "amax");
  if (!at::equal(std::get<0>(mx), want.to(at::kFloat))) { std::fprintf(stderr, "segment_max differs\n"); return 1; }
  // the winner index really attains the maximum
  at::Tensor arg = std::get<1>(mx);
  at::Tensor hit = arg < E;
  at::Tensor picked = msg.gather(0, arg.clamp_max(E - 1));
  if (!at::equal(picked.masked_select(hit), std::get<0>(mx).masked_select(hit))) { std::fprintf(stderr, "argmax differs\n"); return 1; }
  std::printf("ok\n");
  return 0;
}
