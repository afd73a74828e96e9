import sys, time, torch
sys.path.insert(0, '/root/repo')
from gammagl_amd import engine
eng = engine()
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for (E, N) in ((500_000, 50_000), (51_000, 2048), (2_400_000, 170_000)):
    ids = torch.randint(0, N, (E,), generator=g, device=dev)
    full = t(lambda: eng.build_plan(ids, N))
    ro = eng._row_order
    eng._row_order = lambda counts: None
    noro = t(lambda: eng.build_plan(ids, N))
    eng._row_order = ro
    p = eng.build_plan(ids, N)
    order = t(lambda: eng._row_order(p.counts()))
    x = torch.randn(E, 256, device=dev)
    seg = t(lambda: eng._segment_fwd("mean", x, p), reps=50)
    p2 = eng.build_plan(ids, N); p2.row_order = None
    seg2 = t(lambda: eng._segment_fwd("mean", x, p2), reps=50)
    print(f"E={E} N={N}: build_plan {full:.0f} us (without the row order {noro:.0f} us; the order alone {order:.0f} us); segment_mean K=256 {seg:.0f} us, without a row order {seg2:.0f} us")
# the unmodified-reference mini-batch pattern: a FRESH id tensor per call through the public op (plan built per call)
from gammagl_amd import cpp_ops, mpops
C = cpp_ops.load()
for (E, N) in ((500_000, 50_000), (51_000, 2048)):
    x = torch.randn(E, 256, device=dev)
    base = torch.randint(0, N, (E,), generator=g, device=dev)
    def fresh_py():
        ids = base.clone()
        return mpops.unsorted_segment_mean(x, ids, N)
    def fresh_cpp():
        ids = base.clone()
        return C.segment_mean(x, ids, N)
    ids0 = base.clone()
    print(f"E={E} N={N}: unsorted_segment_mean(K=256) on a fresh id tensor per call: python ops {t(fresh_py):.0f} us, "
          f"C++ ops {t(fresh_cpp):.0f} us; on a cached plan {t(lambda: mpops.unsorted_segment_mean(x, ids0, N)):.0f} us")
