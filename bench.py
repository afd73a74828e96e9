#!/usr/bin/env python3
"""bench.py — edges aggregated / second for a 3-layer GCN (hidden 256) training step on an
ogbn-products-sized synthetic graph, MI355X, through the gammagl_amd HIP kernels.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
With N > 1 and no launcher environment (WORLD_SIZE unset) bench.py launches the N ranks ITSELF
(torch.distributed.run, 127.0.0.1, one rank per GPU) and relays rank 0's line; under a launcher it checks
that WORLD_SIZE == --gpus and refuses to print a line for a different world size.  Every rank builds only
its own share of the graph (gammagl_amd.synth.rmat_partitioned): no rank holds the edge list.

* a "step" = one full training step of GCNModel(100 -> 256 -> 256 -> 47, norm='none') on
  precomputed symmetric-normalised edge weights (edge_weight = calc_gcn_norm(edge_index), the
  configuration examples/gcn/gcn_trainer.py:59 sketches) over the whole graph: forward (Linear,
  aggregate, +bias, ReLU, dropout per layer), softmax cross-entropy on the train nodes, backward, Adam
  with weight decay.  GammaGL's GCNConv always computes A (X W) (gcn_conv.py:79): 3 forward CSR SpMMs + 3
  transposed ones per step (--transform-first).  By default a layer whose input is narrower than its output
  computes the same product as (A X) W: the first layer then aggregates 100-wide rows, and its backward needs
  no aggregation (the input features carry no gradient, dW = (A X)^T dH): 5 aggregations per step.
  Same code path for every N (N = 1: no halo exchange);
* value = (aggregations actually executed per step) * E * steps / time over the whole job (max over ranks),
  inputs resident in HBM; "aggregations_per_step" is in the line;
  scaling = "strong": the graph is fixed and node-partitioned over the N GPUs;
* roofline = the dominant kernel (CSR SpMM-sum, feature width 256, forward, rank 0's rows) timed
  with hipEvents on its launch stream: algorithmic bytes E*(4*256+8) + N*(4*256+8) per launch
   / time, against the 8 TB/s HBM3E peak;
* cpu_baseline (N = 1 only) = the reference's own CPU extension (oracle/_ref, compiled from the
  reference sources; our C restatement if it is absent), 1 core (the shipped extension is serial:
  setup.py:50 never defines its OpenMP macro): ONE full-size K=256 aggregate of the benchmark graph itself
  (~20 s), plus the 6 aggregations of a step on a bounded R-MAT sample as a secondary figure.
* GGL_BENCH_EMUL=1 (tests only): gloo + the host-emulated kernels on CPU, to exercise the launcher and the
  N-rank code path where there is no GPU; the line then says "engine": "host-emulation" and is no measurement.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default=os.environ.get("GGL_BENCH_WORKLOAD", "products"),
                   help="products | arxiv | tiny  (node/edge counts of the named dataset, R-MAT)")
    p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--layers", type=int, default=3)
    p.add_argument("--order", default="src", choices=["src", "dst"],
                   help="edge order of the synthetic edge_index (src = coalesced COO as in GammaGL/PyG)")
    p.add_argument("--relabel", default="random", choices=["random", "degree", "none"])
    p.add_argument("--transform-first", action="store_true",
                   help="A (X W) in every layer, as GammaGL's GCNConv writes it (default: a layer whose input is narrower "
                        "than its output computes (A X) W — same product, fewer bytes, no aggregation in layer 1's backward)")
    p.add_argument("--matmul-precision", default="highest", choices=["highest", "high"],
                   help="torch.set_float32_matmul_precision for the dense X W products: 'highest' = IEEE f32 MFMA (the "
                        "line of record); 'high' lets hipBLASLt emulate f32 with bf16 triples on gfx950 (2x faster GEMMs, "
                        "~5e-6 relative error instead of ~8e-7) — reported only as a side figure")
    p.add_argument("--no-tuned-gemm", action="store_true",
                   help="do not load gammagl_amd/tuned/*.csv (PyTorch TunableOp results: which rocBLAS / hipBLASLt f32 "
                        "kernel runs each GEMM shape of the step, chosen offline by tools/tune_gemms.sh)")
    p.add_argument("--no-comparison", action="store_true",
                   help="skip the like-for-like transform-first trainer timed beside the default (profiling runs)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--pmc-traffic", default="auto", choices=["auto", "off"],
                   help="auto: after the timed region (N = 1, products), collect the dominant kernel's HBM-side traffic "
                        "with two rocprofv3 --pmc passes of tools/pmc_probe.py on the same graph (skipped when rocprofv3 "
                        "is not on PATH; falls back to the committed profile)")
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args()


def _ref_or_port():
    from oracle import oracle as orc

    try:
        return orc.load_ref_ext(), "reference", orc
    except Exception:  # noqa: BLE001
        orc.build()
        return None, "port", orc


def cpu_baseline(hidden, classes, seed, full_graph=None):
    """Reference CPU extension (or the oracle port), 1 core.  `full_graph` = (edge_index [2,E] int64 on the
    host, weights [E], N): one K=hidden forward aggregate of the benchmark graph itself; then the 6 aggregations
    of a step on a bounded sample (~8 s)."""
    from gammagl_amd.synth import rmat_graph

    ref, kind, orc = _ref_or_port()
    impl = "reference c_spmm_sum (oracle/_ref)" if kind == "reference" else "oracle C port"
    cores = os.cpu_count() or 1
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(seed)
    out = {}
    if full_graph is not None:
        ei, w, n = full_graph
        E = int(ei.shape[1])
        x = torch.randn(n, hidden, generator=gen)
        t0 = time.perf_counter()
        if ref is not None:
            ref.c_spmm_sum(ei, w, x)
        else:
            orc.spmm_sum_fwd(ei.numpy(), w.numpy(), x.numpy())
        dt = time.perf_counter() - t0
        out = {"value": E / dt, "unit": "edges/s", "cores": 1, "kind": kind,
               "sample": f"ONE forward aggregate (K={hidden}) of the full benchmark graph: N={n}, E={E}, {impl}, "
                         f"{dt:.1f} s on 1 core of {cores}"}
        del x
    n_s, e_s = 400000, (8_000_000 if full_graph is not None else 16_000_000)
    ei = rmat_graph(n_s, e_s, seed=seed + 17, device="cpu")
    E = ei.shape[1]
    w = torch.rand(E, generator=gen)
    widths = [hidden, hidden, classes]
    feats = [torch.randn(n_s, k, generator=gen) for k in widths]
    t0 = time.perf_counter()
    if ref is not None:
        eiT = ei.flip(0).contiguous()
        for x in feats:
            ref.c_spmm_sum(ei, w, x)        # forward aggregate (spmm_sum_cpu_forward)
            ref.c_spmm_sum(eiT, w, x)       # backward = the same loop on the transposed edge list
    else:
        ein, wn = ei.numpy(), w.numpy()
        for x in feats:
            orc.spmm_sum_fwd(ein, wn, x.numpy())
            orc.spmm_sum_bwd(ein, wn, x.numpy())
    dt = time.perf_counter() - t0
    step = {"value": 6 * E / dt, "unit": "edges/s", "cores": 1, "kind": kind,
            "sample": f"R-MAT N={n_s} E={E} (loops incl.), the 6 aggregations of one 3-layer GCN step "
                      f"(K={widths} fwd + transposed bwd), {impl}, {dt:.1f} s on 1 core of {cores}"}
    if out:
        out["step_sample"] = step
    else:
        out = step
    # second baseline: the reference's pure-torch formulation of the same aggregate
    # (mpops/torch.py:16-18,335-342: x[src] * w -> zeros().scatter_add_), all host threads, smaller sample
    torch.set_num_threads(cores)
    e_t = min(E, 1_000_000)
    src, dst, wt = ei[0, :e_t], ei[1, :e_t], w[:e_t]
    t1 = time.perf_counter()
    for x in feats:
        for s_, d_ in ((src, dst), (dst, src)):
            msg = x[s_] * wt.view(-1, 1)
            torch.zeros_like(x).scatter_add_(0, d_.view(-1, 1).expand_as(msg), msg)
    dt_t = time.perf_counter() - t1
    out["torch_fallback"] = {"value": 6 * e_t / dt_t, "unit": "edges/s", "cores": cores,
                             "sample": f"pure-torch mpops formulation (gather * w -> scatter_add_), first {e_t} edges "
                                       f"of the sample, {dt_t:.1f} s on {cores} threads"}
    return out


def measure_traffic(args, hidden, launches=1):
    """HBM-side bytes per launch of the dominant kernel, measured NOW on this box: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section: KiB units, x2 on the read side for gfx950) of
    tools/pmc_probe.py, which rebuilds this run's graph and launches the K=hidden SpMM-sum a few times.  Returns
    (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ggl_pmc_")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(REPO, "tools", "pmc_probe.py"), args.workload, str(hidden), str(args.seed), args.relabel,
                   args.order]
            r = subprocess.run(cmd, cwd=d, env=dict(os.environ, TMPDIR=d), capture_output=True, text=True, timeout=90)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            xs = []
            with open(files[0], newline="") as f:
                for row in csv.DictReader(f):
                    # one wavefront per row (true) for a one-launch K = 256 aggregate, 16 lanes per row (false) for its
                    # 64-column blocks
                    want = "row_reduce_kernel<float, 4, 0, 1, 1, " + ("true" if launches == 1 else "false, 4")
                    if row["Counter_Name"] == counter and want in row["Kernel_Name"]:
                        xs.append(float(row["Counter_Value"]))
            if not xs:
                return None, "kernel not found in the counter file"
            vals[counter] = sum(xs) / len(xs)
        except Exception as ex:  # noqa: BLE001
            return None, f"{type(ex).__name__}: {ex}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py on the same graph " \
        "((2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 read-side correction)"


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(args):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run on this node
    (one per GPU, rendezvous on 127.0.0.1) and relay their output.  Refuses when the node has fewer GPUs."""
    import subprocess

    emul = os.environ.get("GGL_BENCH_EMUL") == "1"
    if not emul:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); not reporting a "
                             f"{args.gpus}-GPU line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    emul = os.environ.get("GGL_BENCH_EMUL") == "1"
    eng = None
    if emul:  # tests only: the launcher + N-rank path on a GPU-less box (gloo, host-emulated kernels)
        import subprocess

        from gammagl_amd import _lib
        from gammagl_amd.ops import Engine

        subprocess.check_call([os.path.join(REPO, "tests", "emul", "build.sh")])
        eng = Engine(_lib.bind(os.path.join(REPO, "tests", "emul", "libggl_emul.so")), require_cuda=False)
        dev, backend = torch.device("cpu"), "gloo"
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
        torch.cuda.set_device(local_rank)
        dev, backend = torch.device("cuda", local_rank), "nccl"
    if world > 1:
        import torch.distributed as dist

        if emul:
            dist.init_process_group(backend)
        else:
            dist.init_process_group(backend, device_id=dev)

    from gammagl_amd.dist import run_distributed_bench
    from gammagl_amd.synth import DATASETS

    torch.set_float32_matmul_precision(args.matmul_precision)
    tuned = None
    if not emul and not args.no_tuned_gemm and args.matmul_precision == "highest":
        # library GEMM selection only: the same IEEE f32 products, each shape on the rocBLAS / hipBLASLt kernel an offline
        # TunableOp pass measured fastest on this part (the file's validators — torch / HIP / library versions, gfx
        # arch — must match, otherwise torch ignores it and the default heuristics pick)
        path = os.path.join(REPO, "gammagl_amd", "tuned", f"tunableop_gfx950_{args.workload}.csv")
        if os.path.exists(path):
            try:
                import torch.cuda.tunable as tun

                tun.enable(True)
                tun.tuning_enable(False)
                tuned = os.path.relpath(path, REPO) if tun.read_file(path) else None
                if tuned is None:
                    tun.enable(False)
            except Exception:  # noqa: BLE001
                tuned = None

    if args.workload == "tiny":
        n_nodes, n_edges, f_in, n_cls = 20000, 400000, 100, 47
    else:
        n_nodes, n_edges, f_in, n_cls = DATASETS[args.workload]
    out, pg = run_distributed_bench(args, dev, rank, world, n_nodes, n_edges, f_in, n_cls, eng=eng)
    out["engine"] = "host-emulation (launcher test, not a measurement)" if emul else "hip"
    out["config"]["tuned_gemm_selection"] = tuned
    out["config"]["matmul_precision"] = args.matmul_precision + (" (IEEE f32)" if args.matmul_precision == "highest"
                                                                   else " (hipBLASLt f32 emulated with bf16 triples: NOT the line of record)")
    if emul:
        out["roofline"] = None
    if world > 1:
        import torch.distributed as dist

        assert dist.get_world_size() == args.gpus
    if rank == 0:
        if world == 1 and not emul and args.pmc_traffic == "auto" and args.workload == "products":
            t, src = measure_traffic(args, args.hidden, int(out["roofline"].get("launches_per_aggregate", 1)))
            if t is not None:
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = t, src
            elif out["roofline"].get("traffic") is not None:
                out["roofline"]["traffic_source"] += f" [in-run collection unavailable: {src}]"
        rf = out.get("roofline") or {}
        if rf.get("traffic") and rf.get("ms_per_launch"):
            # the HBM side of the same launch: measured bytes / its duration against the peak (frac above is algorithmic:
            # it counts the gathers that L2 served as if they had crossed the fabric)
            rf["traffic_GBps"] = rf["traffic"] / (rf["ms_per_launch"] * 1e-3) / 1e9
            rf["traffic_frac"] = rf["traffic_GBps"] / rf["peak"]
        if world == 1 and not args.no_cpu_baseline and not emul:
            # the benchmark graph itself on the host (rank 0 holds all of it at N = 1)
            ei = torch.cat([torch.stack([pg.ei_loc[0] + pg.lo, pg.ei_loc[1] + pg.lo]).cpu()], dim=1)
            full = (ei.contiguous(), pg.w_loc.cpu(), n_nodes) if args.workload != "tiny" else None
            out["cpu_baseline"] = cpu_baseline(args.hidden, n_cls, args.seed, full)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
