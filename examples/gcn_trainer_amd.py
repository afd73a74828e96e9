#!/usr/bin/env python3
"""examples/gcn/gcn_trainer.py of GammaGL, on the MI355X backend.

Same flow and flags as the reference trainer (examples/gcn/gcn_trainer.py:51-142): add self-loops once,
GCNModel(feature_dim, hidden_dim, num_class, drop_rate, num_layers, norm), Adam(lr, weight_decay=l2_coef),
softmax cross-entropy on the train nodes, accuracy on val/test.  Datasets cannot be downloaded here, so
without --data it trains on a seeded Cora-sized homophilous synthetic graph (labels recoverable from
features + neighbourhood); --data points at an .npz with x, y, edge_index, train_idx, val_idx, test_idx.

    python examples/gcn_trainer_amd.py --n_epoch 50 --hidden_dim 16
    python examples/gcn_trainer_amd.py --gpu -1 --n_epoch 20       # BASELINE config 1: TL_BACKEND=torch on the CPU

With --gpu -1 (gcn_trainer.py's own flag for "no GPU") every tensor stays on the host and the same ops dispatch to the
host build of the kernel sources (CPU dispatch key, libggl_mpops_host.so) — the plumbing configuration of BASELINE.json.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd.layers import GCNModel, add_self_loops  # noqa: E402
from gammagl_amd.synth import homophilous_graph  # noqa: E402


def load(args, dev):
    if args.data:
        d = np.load(args.data)
        t = lambda k, dt: torch.as_tensor(d[k], dtype=dt, device=dev)  # noqa: E731
        return (t("x", torch.float32), t("y", torch.int64), t("edge_index", torch.int64), t("train_idx", torch.int64),
                t("val_idx", torch.int64), t("test_idx", torch.int64))
    n, f, c = 2708, 1433, 7   # Cora's node / feature / class counts
    x, y, ei = homophilous_graph(n, f, c, deg=2, seed=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    perm = torch.randperm(n, generator=g, device=dev)
    return x, y, ei, perm[:140], perm[140:640], perm[640:1640]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--n_epoch", type=int, default=200)
    p.add_argument("--hidden_dim", type=int, default=16)
    p.add_argument("--drop_rate", type=float, default=0.5)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--norm", type=str, default="both")
    p.add_argument("--l2_coef", type=float, default=5e-4)
    p.add_argument("--data", type=str, default="")
    p.add_argument("--gpu", type=int, default=0)
    args = p.parse_args()
    dev = torch.device("cuda", args.gpu) if args.gpu >= 0 else torch.device("cpu")
    x, y, edge_index, train_idx, val_idx, test_idx = load(args, dev)
    n = x.shape[0]
    edge_index = add_self_loops(edge_index, n)                       # gcn_trainer.py:58
    torch.manual_seed(0)
    net = GCNModel(x.shape[1], args.hidden_dim, int(y.max()) + 1, args.drop_rate, args.num_layers, args.norm).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, weight_decay=args.l2_coef)
    best_val, best_state = 0.0, None
    for epoch in range(args.n_epoch):
        net.train()
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(net(x, edge_index, None, n)[train_idx], y[train_idx])
        loss.backward()
        opt.step()
        net.eval()
        with torch.no_grad():
            logits = net(x, edge_index, None, n)
        val_acc = float((logits[val_idx].argmax(1) == y[val_idx]).float().mean())
        if epoch % 10 == 0 or epoch == args.n_epoch - 1:
            print("Epoch [{:0>3d}]   train loss: {:.4f}  val acc: {:.4f}".format(epoch + 1, float(loss.detach()), val_acc))
        if val_acc > best_val:
            best_val, best_state = val_acc, {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(best_state)
    net.eval()
    with torch.no_grad():
        logits = net(x, edge_index, None, n)
    print("Test acc:  {:.4f}".format(float((logits[test_idx].argmax(1) == y[test_idx]).float().mean())))


if __name__ == "__main__":
    main()
