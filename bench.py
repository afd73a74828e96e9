#!/usr/bin/env python3
"""bench.py — edges aggregated / second for a 3-layer GCN (hidden 256) training step on an
ogbn-products-sized synthetic graph, MI355X, through the gammagl_amd HIP kernels.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.

* a "step" = one full training step of GCNModel(100 -> 256 -> 256 -> 47, norm='none') on
  precomputed symmetric-normalised edge weights (edge_weight = calc_gcn_norm(edge_index), the
  configuration examples/gcn/gcn_trainer.py:59 sketches) over the whole graph: forward (Linear ->
  aggregate -> +bias -> ReLU -> dropout per layer), softmax cross-entropy on the train nodes,
  backward, Adam with weight decay.  Every step aggregates 6 x E edges: 3 forward CSR SpMMs + 3
  transposed SpMMs in backward.  Same code path for every N (N = 1: no halo exchange);
* value = 6 * E * steps / time over the whole job (max over ranks), inputs resident in HBM;
  scaling = "strong": the graph is fixed and node-partitioned over the N GPUs;
* roofline = the dominant kernel (CSR SpMM-sum, feature width 256, forward, rank 0's rows) timed
  with hipEvents on its launch stream: algorithmic bytes E*(4*256+8) + N*(4*256+8) per launch
   / time, against the 8 TB/s HBM3E peak;
* cpu_baseline (N = 1 only) = the reference's own CPU extension (oracle/_ref, compiled from the
  reference sources; our C restatement if it is absent) running the same 6 aggregations on a bounded
  R-MAT sample, 1 core (the shipped extension is serial: setup.py:50 never defines its OpenMP macro).
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
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args()


def cpu_baseline(hidden, classes, seed):
    """Reference CPU extension (or the oracle port) on a bounded sample: ~10-20 s of one core."""
    from gammagl_amd.synth import rmat_graph
    from oracle import oracle as orc

    n_s, e_s = 400000, 16_000_000
    ei = rmat_graph(n_s, e_s, seed=seed + 17, device="cpu")
    E = ei.shape[1]
    gen = torch.Generator().manual_seed(seed)
    w = torch.rand(E, generator=gen)
    widths = [hidden, hidden, classes]
    feats = [torch.randn(n_s, k, generator=gen) for k in widths]
    ref = None
    try:
        ref = orc.load_ref_ext()
    except Exception:  # noqa: BLE001
        ref = None
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    if ref is not None:
        kind = "reference"
        eiT = ei.flip(0).contiguous()
        for x in feats:
            ref.c_spmm_sum(ei, w, x)        # forward aggregate (spmm_sum_cpu_forward)
            ref.c_spmm_sum(eiT, w, x)       # backward = the same loop on the transposed edge list
    else:
        kind = "port"
        orc.build()
        ein, wn = ei.numpy(), w.numpy()
        for x in feats:
            orc.spmm_sum_fwd(ein, wn, x.numpy())
            orc.spmm_sum_bwd(ein, wn, x.numpy())
    dt = time.perf_counter() - t0
    impl = "reference c_spmm_sum (oracle/_ref)" if kind == "reference" else "oracle C port"
    # second baseline: the reference's pure-torch formulation of the same aggregate
    # (mpops/torch.py:16-18,335-342: x[src] * w -> zeros().scatter_add_), all host threads, smaller sample
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    e_t = min(E, 3_000_000)
    src, dst, wt = ei[0, :e_t], ei[1, :e_t], w[:e_t]
    t1 = time.perf_counter()
    for x in feats:
        for s_, d_ in ((src, dst), (dst, src)):
            msg = x[s_] * wt.view(-1, 1)
            torch.zeros_like(x).scatter_add_(0, d_.view(-1, 1).expand_as(msg), msg)
    dt_t = time.perf_counter() - t1
    return {
        "value": 6 * E / dt, "unit": "edges/s", "cores": 1, "kind": kind,
        "sample": f"R-MAT N={n_s} E={E} (loops incl.), the 6 aggregations of one 3-layer GCN step "
                  f"(K={widths} fwd + transposed bwd), {impl}, {dt:.1f} s on 1 core of {cores}",
        "torch_fallback": {"value": 6 * e_t / dt_t, "unit": "edges/s", "cores": cores,
                           "sample": f"pure-torch mpops formulation (gather * w -> scatter_add_), first {e_t} edges of "
                                     f"the same sample, {dt_t:.1f} s on {cores} threads"},
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    from gammagl_amd.dist import run_distributed_bench
    from gammagl_amd.synth import DATASETS

    if args.workload == "tiny":
        n_nodes, n_edges, f_in, n_cls = 20000, 400000, 100, 47
    else:
        n_nodes, n_edges, f_in, n_cls = DATASETS[args.workload]
    out = run_distributed_bench(args, dev, rank, world, n_nodes, n_edges, f_in, n_cls)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.hidden, n_cls, args.seed)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
