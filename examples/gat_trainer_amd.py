#!/usr/bin/env python3
"""examples/gat/gat_trainer.py of GammaGL on the MI355X backend (fused edge-softmax + aggregate kernels).

Same flow and flags as the reference trainer: GATModel(feature_dim, hidden_dim, num_class, heads, drop_rate,
num_layers), self-loops added once, Adam with weight decay, cross-entropy on the train nodes.  Trains on a
seeded Cora-sized homophilous synthetic graph (datasets cannot be downloaded here).

    python examples/gat_trainer_amd.py --n_epoch 100 [--unfused]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.layers import GATModel, add_self_loops  # noqa: E402
from gammagl_amd.synth import homophilous_graph  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--lr", type=float, default=0.005)
    p.add_argument("--n_epoch", type=int, default=200)
    p.add_argument("--hidden_dim", type=int, default=8)
    p.add_argument("--heads", type=int, default=8)
    p.add_argument("--drop_rate", type=float, default=0.6)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--l2_coef", type=float, default=5e-4)
    p.add_argument("--unfused", action="store_true", help="GATConv on the segment ops instead of FusedGATConv")
    p.add_argument("--gpu", type=int, default=0)
    args = p.parse_args()
    dev = torch.device("cuda", args.gpu) if args.gpu >= 0 else torch.device("cpu")   # gat_trainer.py's own "--gpu -1"
    n, f, c = 2708, 1433, 7
    x, y, edge_index = homophilous_graph(n, f, c, deg=2, seed=0, device=dev)
    edge_index = add_self_loops(edge_index, n)
    perm = torch.randperm(n, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    train_idx, val_idx, test_idx = perm[:140], perm[140:640], perm[640:1640]
    torch.manual_seed(0)
    net = GATModel(f, args.hidden_dim, c, args.heads, args.drop_rate, args.num_layers, fused=not args.unfused).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, weight_decay=args.l2_coef)
    best_val, best_state = 0.0, None
    for epoch in range(args.n_epoch):
        net.train()
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(net(x, edge_index, n)[train_idx], y[train_idx])
        loss.backward()
        opt.step()
        net.eval()
        with torch.no_grad():
            logits = net(x, edge_index, n)
        val_acc = float((logits[val_idx].argmax(1) == y[val_idx]).float().mean())
        if epoch % 10 == 0 or epoch == args.n_epoch - 1:
            print("Epoch [{:0>3d}]   train loss: {:.4f}  val acc: {:.4f}".format(epoch + 1, float(loss.detach()), val_acc))
        if val_acc > best_val:
            best_val, best_state = val_acc, {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(best_state)
    net.eval()
    with torch.no_grad():
        logits = net(x, edge_index, n)
    print("Test acc:  {:.4f}".format(float((logits[test_idx].argmax(1) == y[test_idx]).float().mean())))


if __name__ == "__main__":
    main()
