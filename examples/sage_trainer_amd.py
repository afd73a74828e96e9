#!/usr/bin/env python3
"""examples/graphsage/reddit_sage_trainer.py of GammaGL on the MI355X backend: neighbour-sampled mini-batch
GraphSAGE — NeighborSampler(sample_lists=[25, 10], batch_size=2048) on the device, GraphSAGE_Sample_Model,
Adam — on a seeded homophilous synthetic graph (Reddit cannot be downloaded here).  With torchrun every rank
is a replica that samples its own share of each batch; gradients are all-reduced (RCCL).

    python examples/sage_trainer_amd.py --n_epoch 3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/sage_trainer_amd.py
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.sampler import BlockSampler, NeighborSampler  # noqa: E402
from gammagl_amd.synth import homophilous_graph  # noqa: E402
from gammagl_amd.trainer import SAGEBlockTrainer, SAGETrainer  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--lr", type=float, default=0.005)
    p.add_argument("--n_epoch", type=int, default=3)
    p.add_argument("--hidden_dim", type=int, default=256)
    p.add_argument("--drop_rate", type=float, default=0.2)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--batch_size", type=int, default=2048)
    p.add_argument("--nodes", type=int, default=200_000)
    p.add_argument("--sampler", default="static", choices=["static", "dynamic"],
                   help="static: BlockSampler (fixed capacities, no host reads; one replayed hipGraph per batch on a "
                        "single GPU); dynamic: NeighborSampler (exact shapes, two host reads per hop)")
    p.add_argument("--gpu", type=int, default=0, help="-1: everything on the host (the reference trainer's own flag; CPU "
                                                      "tensors dispatch to the host build of the kernel sources)")
    args = p.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    on_gpu = args.gpu >= 0
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", str(args.gpu)))) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        if on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    n, f, c = args.nodes, 128, 16
    x, y, edge_index = homophilous_graph(n, f, c, deg=10, p_same=0.7, signal=0.3, seed=0, device=dev)
    perm = torch.randperm(n, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    train_idx, test_idx = perm[: n // 2], perm[n // 2: n // 2 + 20000]
    sampler = NeighborSampler(edge_index, [25, 10], num_nodes=n)       # reddit_sage_trainer.py:55-57
    per_rank = args.batch_size // world
    graphed = None
    if args.sampler == "static":
        bs = BlockSampler(edge_index, [25, 10], num_nodes=n)
        caps = bs.calibrate(per_rank, trials=8, slack=1.3)
        tr = SAGEBlockTrainer(bs, f, args.hidden_dim, c, num_layers=args.num_layers, drop_rate=args.drop_rate,
                              lr=args.lr, device=dev, caps=caps, world=world)
        if world == 1 and on_gpu:   # the whole step — sampling included — as one replayed hipGraph
            seed_buf = train_idx[:per_rank].clone()
            graphed = tr.capture(x, y, seed_buf)
    else:
        tr = SAGETrainer(sampler, f, args.hidden_dim, c, num_layers=args.num_layers, drop_rate=args.drop_rate,
                         lr=args.lr, device=dev, world=world)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    for epoch in range(args.n_epoch):
        order = train_idx[torch.randperm(train_idx.numel(), generator=g, device=dev)]
        n_batches = order.numel() // args.batch_size
        for b in range(n_batches):
            seeds = order[b * args.batch_size + rank * per_rank: b * args.batch_size + (rank + 1) * per_rank]
            if graphed is not None:
                seed_buf.copy_(seeds)
                loss = tr.replay()
            else:
                loss = tr.step(x, y, seeds)
            if rank == 0 and b % 10 == 0:
                print("Epoch [{:0>3d}] batch {:3d}/{}  train loss: {:.4f}".format(epoch + 1, b, n_batches, float(loss.detach())))
        tr.net.eval()
        with torch.no_grad():
            dst, n_id, adjs = sampler.sample(test_idx[:4096])
            acc = float((tr.net(x[n_id], adjs).argmax(1) == y[dst]).float().mean())
        if rank == 0:
            print("Epoch [{:0>3d}] sampled test acc: {:.4f}".format(epoch + 1, acc))
        if args.sampler == "static" and bs.overflow_count():
            raise RuntimeError("a sampled block exceeded its calibrated capacity: re-run with a larger slack")
    # layer-wise full-neighbourhood inference over all nodes (GraphSAGE_Sample_Model.inference)
    full = NeighborSampler(edge_index, [-1], num_nodes=n)
    logits = tr.net.inference(x, full)
    if rank == 0:
        print("Full-neighbourhood test acc: {:.4f}".format(float((logits[test_idx].argmax(1) == y[test_idx]).float().mean())))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
