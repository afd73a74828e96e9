"""Host cost of ONE op call through the three routes to the same kernel, on a graph small enough that the kernel is
a few microseconds: the ctypes engine called directly, the Python-registered dispatcher op
(torch.ops.gammagl_amd.*: dispatcher -> Python -> ctypes), and the C++-registered one (torch.ops.ggl.*: dispatcher ->
C++ -> C ABI).  Forward under no_grad, and forward + backward.  Wall clock per call over `reps` calls, device
synchronised once at the end (the launches queue; what is timed is how fast the host can issue them)."""
import argparse
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--reps", type=int, default=3000)
    ap.add_argument("--nodes", type=int, default=2000)
    ap.add_argument("--edges", type=int, default=20000)
    ap.add_argument("--width", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device(a.device)
    import gammagl_amd
    from gammagl_amd import cpp_ops, torch_ops

    g = torch.Generator(device=dev).manual_seed(0)
    ei = torch.randint(0, a.nodes, (2, a.edges), generator=g, device=dev)
    w = torch.rand(a.edges, generator=g, device=dev)
    x = torch.randn(a.nodes, a.width, generator=g, device=dev)
    msg = torch.randn(a.edges, a.width, generator=g, device=dev)
    eng = gammagl_amd.engine(x)
    P, C = torch_ops.ops, cpp_ops.load()
    routes = {
        "engine (ctypes, direct)": (lambda t: eng.c_spmm_sum(ei, w, t), lambda t: eng.c_segment_sum(t, ei[1], a.nodes)),
        "torch.ops.gammagl_amd (Python-registered)": (lambda t: P.spmm_sum(ei, w, t), lambda t: P.segment_sum(t, ei[1], a.nodes)),
        "torch.ops.ggl (C++-registered)": (lambda t: C.spmm_sum(ei, w, t), lambda t: C.segment_sum(t, ei[1], a.nodes)),
    }

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def timed(fn):
        for _ in range(50):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        sync()
        return (time.perf_counter() - t0) / a.reps * 1e6

    print(f"# {dev} N={a.nodes} E={a.edges} K={a.width}, {a.reps} calls each; microseconds per call")
    print(f"{'route':45s} {'spmm fwd':>10s} {'spmm f+b':>10s} {'segsum fwd':>11s} {'segsum f+b':>11s}")
    dst = ei[1].contiguous()
    for name, (spmm, seg) in routes.items():
        def fwd(f, t):
            with torch.no_grad():
                f(t)

        def fb(f, t):
            tt = t.detach().requires_grad_(True)
            f(tt).sum().backward()

        res = [timed(lambda: fwd(spmm, x)), timed(lambda: fb(spmm, x)), timed(lambda: fwd(seg, msg)), timed(lambda: fb(seg, msg))]
        print(f"{name:45s} " + " ".join(f"{v:10.1f}" for v in res))
    # the fused route (round 4: registered from C++ as well)
    H, Cc = 8, 8
    xg = torch.randn(a.nodes, H, Cc, generator=g, device=dev)
    el, er = torch.randn(a.nodes, H, generator=g, device=dev), torch.randn(a.nodes, H, generator=g, device=dev)
    act = torch.randn(a.nodes, a.width, generator=g, device=dev)
    bias = torch.randn(a.width, generator=g, device=dev)
    gp = eng.graph_plan(ei, a.nodes)
    fused = {
        "engine (ctypes, direct)": (lambda t: eng.gat_fused(ei, el, er, t, 0.2, a.nodes, 0.0, True),
                                    lambda t: eng.bias_act(t, bias, True, 0.0, True),
                                    lambda t: eng.spmm_epi(gp, w, t, "sum", None, bias, True, 0.0, True)),
        "torch.ops.gammagl_amd (Python-registered)": (lambda t: P.gat_fused(ei, el, er, t, 0.2, a.nodes, 0.0),
                                                      lambda t: P.bias_act(t, bias, True, 0.0), None),
        "torch.ops.ggl (C++-registered)": (lambda t: C.gat_fused(ei, el, er, t, 0.2, a.nodes, 0.0),
                                           lambda t: C.bias_act(t, bias, True, 0.0),
                                           lambda t: C.spmm_epi(ei, w, t, False, None, bias, True, 0.0)),
    }
    print(f"\n{'route':45s} {'gat fwd':>10s} {'gat f+b':>10s} {'biasact fwd':>11s} {'biasact f+b':>11s} {'epi fwd':>10s} {'epi f+b':>10s}")
    for name, (gat, ba, epi) in fused.items():
        def fwd(f, t):
            with torch.no_grad():
                f(t)

        def fb(f, t):
            tt = t.detach().requires_grad_(True)
            f(tt).sum().backward()

        res = [timed(lambda: fwd(gat, xg)), timed(lambda: fb(gat, xg)), timed(lambda: fwd(ba, act)), timed(lambda: fb(ba, act))]
        res += [timed(lambda: fwd(epi, x)), timed(lambda: fb(epi, x))] if epi is not None else [float("nan")] * 2
        print(f"{name:45s} " + " ".join(f"{v:10.1f}" for v in res))


if __name__ == "__main__":
    main()
