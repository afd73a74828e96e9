"""Round-5 A/B probe on the products-sized graph: (1) K = 256 SpMM-sum aggregate with the hub walk once per aggregate
(hub_one_launch = 1) vs once per 64-column block (0), random and degree node order; (2) gspmm max forward + backward with the
backward through int64 witnesses / int32 witnesses / the 1-bit winner mask, K = 64 / 256, and the mask pre-pass alone."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine
from gammagl_amd.layers import calc_gcn_norm
from gammagl_amd.synth import DATASETS, rmat_graph
dev = torch.device("cuda", 0); eng = engine()
what = sys.argv[1:] or ["hub", "max"]
n, e, _, _ = DATASETS["products"]

def ev(fn, reps=8):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for relabel in (("random", "degree") if "hub" in what else ("random",)):
    ei = rmat_graph(n, e, seed=0, device=dev, relabel=relabel) if relabel != "random" else rmat_graph(n, e, seed=0, device=dev)
    E = ei.shape[1]
    w = calc_gcn_norm(ei, n).contiguous()
    gp = eng.graph_plan(ei, n)
    if "hub" in what:
        x = torch.randn(n, 256, device=dev)
        with torch.no_grad():
            eng.c_spmm_sum(ei, w, x)
            for one in (1, 0, 1, 0):
                eng.set_option("hub_one_launch", one)
                print(f"[{relabel}] spmm sum K=256 hub_one_launch={one}: {ev(lambda: eng.c_spmm_sum(ei, w, x)):7.3f} ms", flush=True)
            eng.set_option("hub_one_launch", 1)
            eng.set_option("exact_long_rows", 0)
            print(f"[{relabel}] spmm sum K=256 chunked walk        : {ev(lambda: eng.c_spmm_sum(ei, w, x)):7.3f} ms", flush=True)
            eng.set_option("exact_long_rows", 1)
            for K in (64, 128):
                xk = torch.randn(n, K, device=dev)
                print(f"[{relabel}] spmm sum K={K}: {ev(lambda: eng.c_spmm_sum(ei, w, xk)):7.3f} ms", flush=True)
        del x
    if "max" in what and relabel == "random":
        for K in (64, 128, 256):
            xk = torch.randn(n, K, device=dev, requires_grad=True)
            go = torch.randn(n, K, device=dev)
            def fb():
                xk.grad = None
                eng.c_spmm_max(ei, w, xk).backward(go)
            with torch.no_grad():
                f = ev(lambda: eng.c_spmm_max(ei, w, xk))
            line = f"gspmm max K={K:3d}: fwd {f:7.3f}"
            forms = (("int64 witnesses", dict(maxbwd_mask=0, maxbwd_arg32=0)), ("int32 witnesses", dict(maxbwd_mask=0, maxbwd_arg32=1)),
                     ("mask fwd-order (select)", dict(maxbwd_mask=1)), ("mask fwd-order (writelane)", dict(maxbwd_mask=1, maxbwd_mask_wlane=1)),
                     ("mask scattered", dict(maxbwd_mask=1, maxbwd_mask_scatter=1)))
            for name, opts in forms:
                for k in ("maxbwd_mask", "maxbwd_arg32", "maxbwd_mask_wlane", "maxbwd_mask_scatter"):
                    eng.set_option(k, opts.get(k, 0))
                line += f" | fwd+bwd {name} {ev(fb):7.3f}"
            eng.set_option("maxbwd_mask", 128); eng.set_option("maxbwd_arg32", 1)
            eng.set_option("maxbwd_mask_wlane", 1); eng.set_option("maxbwd_mask_scatter", 0)
            print(line, flush=True)
            # the mask pre-pass alone, each form
            with torch.no_grad():
                _, arg = eng._spmm_fwd("max", gp.fwd, gp.col, w, xk.detach(), n)
                L = eng.lib
                mask = torch.empty(int(L.ggl_spmm_max_mask_bytes(E, K)) // 4 + 4, dtype=torch.int32, device=dev)
                fs = gp.fwd.c_struct(None)
                tp = gp.tpos
                st = eng._stream(dev)
                for name, tpp, wl in (("fwd-order select", None, 0), ("fwd-order writelane", None, 1), ("scattered", tp, 0)):
                    eng.set_option("maxbwd_mask_wlane", wl)
                    t = ev(lambda: eng._check(L.ggl_spmm_max_mask(ctypes.byref(fs), ctypes.c_void_p(gp.col.data_ptr()),
                                                                  ctypes.c_void_p(tpp.data_ptr()) if tpp is not None else None,
                                                                  ctypes.c_void_p(arg.data_ptr()), K, ctypes.c_void_p(mask.data_ptr()), st)))
                    print(f"   mask pre-pass K={K} {name}: {t:7.3f} ms", flush=True)
                eng.set_option("maxbwd_mask_wlane", 1)
            del xk, go, arg, mask
    eng.clear_caches(); del gp, ei, w
    torch.cuda.empty_cache()
