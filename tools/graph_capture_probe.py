#!/usr/bin/env python3
"""Can a whole GCN training step (our HIP ops + hipBLASLt + Adam) be captured in a hipGraph?
The C ABI never syncs on the hot path and allocates nothing itself, so it should be; this probe
checks it on Cora- and arxiv-sized graphs and times eager vs graph replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gammagl_amd import engine  # noqa: E402
from gammagl_amd.dist import DistGCN, PartitionedGraph  # noqa: E402
from gammagl_amd.layers import calc_gcn_norm  # noqa: E402
from gammagl_amd.synth import rmat_graph  # noqa: E402
import torch.nn.functional as F  # noqa: E402

dev = torch.device("cuda", 0)
eng = engine()
for name, (n, e, f, c, hid, layers) in {"cora": (2708, 10556, 1433, 7, 16, 2), "arxiv": (169343, 2315598, 128, 40, 256, 3)}.items():
    ei = rmat_graph(n, e, seed=0, device=dev)
    w = calc_gcn_norm(ei, n).contiguous()
    pg = PartitionedGraph(ei, w, n, 0, 1, eng=eng)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, f, generator=g, device=dev)
    y = torch.randint(0, c, (n,), generator=g, device=dev)
    idx = torch.arange(0, n, 2, device=dev)
    torch.manual_seed(0)
    net = DistGCN(f, hid, c, layers, drop_rate=0.0).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
    loss_buf = torch.zeros((), device=dev)

    def step():
        opt.zero_grad(set_to_none=False)
        loss = F.cross_entropy(net(x, pg)[idx], y[idx])
        loss.backward()
        net.join()  # side-stream weight gradients back into the (captured) main stream
        opt.step()
        loss_buf.copy_(loss.detach())

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    l_eager = float(loss_buf)
    try:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gr.replay()
        torch.cuda.synchronize()
        graphed = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{name}: N={n} E={ei.shape[1]} eager {eager:.3f} ms/step (loss {l_eager:.4f}), hipGraph replay {graphed:.3f} ms/step "
              f"(loss {float(loss_buf):.4f}) -> {eager / graphed:.2f}x")
    except Exception as ex:  # noqa: BLE001
        print(f"{name}: eager {eager:.3f} ms/step; graph capture FAILED: {type(ex).__name__}: {str(ex)[:300]}")
