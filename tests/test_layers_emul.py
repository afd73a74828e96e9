"""gammagl_amd/layers.py without a GPU: in a subprocess the package's engine is replaced by the host-emulation
build (test-only injection, as in the gloo tests) and the layer classes are checked against the formulas of the
reference layers written out in plain torch: GCNConv (all norms, cached norm weights, padded class widths, fused
epilogue through GCNModel, learnable edge weights), SAGEConv (mean / gcn / pool / lstm, fused and segment routes),
GATConv == FusedGATConv (incl. 41 channels per head), the GAT / GraphSAGE models."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _body():
    import torch

    sys.path.insert(0, REPO)
    import gammagl_amd
    from gammagl_amd import _lib, torch_ops
    from gammagl_amd.ops import Engine

    eng = Engine(_lib.bind(os.path.join(HERE, "emul", "libggl_emul.so")), require_cuda=False)
    gammagl_amd._engine = eng
    torch_ops.register_backend(lambda: eng, "CPU")
    from gammagl_amd import layers
    from gammagl_amd.sampler import NeighborSampler

    g = torch.Generator().manual_seed(0)
    N, E = 60, 700
    ei = torch.randint(0, N, (2, E), generator=g)
    ei = layers.add_self_loops(ei, N)
    x = torch.randn(N, 10, generator=g)
    src, dst = ei[0], ei[1]

    def agg(h, w):  # sum_{j->i} w_ij h_j
        return torch.zeros(N, h.shape[1]).index_add_(0, dst, h[src] * w.unsqueeze(1))

    deg_s = torch.bincount(src, minlength=N).float()
    deg_d = torch.bincount(dst, minlength=N).float()
    for norm, w in (("both", deg_s.pow(-0.5)[src] * deg_d.pow(-0.5)[dst]), ("left", (1 / deg_s)[src]),
                    ("right", (1 / deg_d)[dst]), ("none", torch.ones(ei.shape[1]))):
        for out_c in (16, 10, 7):   # multiple of 4 (fused store), padded 10 -> 12, narrow (propagate route)
            conv = layers.GCNConv(10, out_c, norm=norm)
            torch.nn.init.normal_(conv.bias)
            xa = x.clone().requires_grad_(True)
            ya = conv(xa, ei)
            ref = agg(xa @ conv.linear.weight.t(), w) + conv.bias
            torch.testing.assert_close(ya, ref, rtol=1e-5, atol=1e-5)
            assert ya.shape == (N, out_c)
            gref = torch.autograd.grad(ref.square().sum(), (xa, conv.linear.weight, conv.bias))
            ggot = torch.autograd.grad(ya.square().sum(), (xa, conv.linear.weight, conv.bias))
            for a, b in zip(ggot, gref):
                torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
            y2 = conv(x, ei)  # second call: norm weights come from the cache on the plan
            torch.testing.assert_close(y2, ya.detach(), rtol=0, atol=0)
    # learnable edge weight keeps its gradient
    ew = torch.rand(ei.shape[1], generator=g, requires_grad=True)
    conv = layers.GCNConv(10, 8, norm="none")
    conv(x, ei, ew).sum().backward()
    torch.testing.assert_close(ew.grad, (x @ conv.linear.weight.t())[src].sum(1), rtol=1e-5, atol=1e-5)
    # GCNModel: eval mode == conv -> relu -> conv written out; train mode drops units
    net = layers.GCNModel(10, 16, 5, drop_rate=0.5, num_layers=2)
    net.eval()
    w_both = deg_s.pow(-0.5)[src] * deg_d.pow(-0.5)[dst]
    h = torch.relu(agg(x @ net.conv[0].linear.weight.t(), w_both) + net.conv[0].bias)
    ref = agg(h @ net.conv[1].linear.weight.t(), w_both) + net.conv[1].bias
    torch.testing.assert_close(net(x, ei, None, N), ref, rtol=1e-5, atol=1e-5)
    net.train()
    assert not torch.equal(net(x, ei, None, N), net(x, ei, None, N))
    # SAGEConv: aggregators vs formulas, segment route and fused route
    nd = 25
    blk = ei[:, ei[1] < nd]
    cnt = torch.bincount(blk[1], minlength=nd).clamp(min=1).unsqueeze(1)
    for thr in (10**12, 0):
        layers.FUSED_MIN_EDGES = thr
        sage = layers.SAGEConv(10, 6, aggr="mean")
        hs = sage.fc_neigh(x)
        ref = torch.zeros(nd, 6).index_add_(0, blk[1], hs[blk[0]]) / cnt + sage.fc_self(x[:nd]) + sage.bias
        torch.testing.assert_close(sage((x, x[:nd]), blk), ref, rtol=1e-5, atol=1e-5)
    # "+ fc_self(x_dst) + bias -> act" fused into the aggregate's store == the three-pass form, bit for bit,
    # values and gradients, on both routes (segment route for sampled blocks, fused SpMM-mean for big lists)
    for thr in (10**12, 0):
        layers.FUSED_MIN_EDGES = thr
        for act in (torch.relu, None):
            sage = layers.SAGEConv(10, 8, activation=act, aggr="mean")
            torch.nn.init.normal_(sage.bias)
            outs = []
            for fuse in (True, False):
                layers.SAGE_FUSE_EPILOGUE = fuse
                xa = x.clone().requires_grad_(True)
                y = sage((xa, xa[:nd]), blk)
                gr = torch.autograd.grad(y.square().sum(), [xa] + list(sage.parameters()))
                outs.append([y.detach()] + list(gr))
            for a, b in zip(*outs):
                torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
            assert torch.equal(outs[0][0], outs[1][0])
    layers.SAGE_FUSE_EPILOGUE = True
    # input narrower than output (10 -> 16): the layer aggregates first, on both routes == the formula
    for thr in (10**12, 0):
        layers.FUSED_MIN_EDGES = thr
        sage = layers.SAGEConv(10, 16, activation=torch.relu, aggr="mean")
        torch.nn.init.normal_(sage.bias)
        xa = x.clone().requires_grad_(True)
        y = sage((xa, xa[:nd]), blk)
        ref = torch.relu((torch.zeros(nd, 10).index_add_(0, blk[1], x[blk[0]]) / cnt) @ sage.fc_neigh.weight.t()
                         + x[:nd] @ sage.fc_self.weight.t() + sage.bias)
        torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
        gr = torch.autograd.grad(y.square().sum(), [xa] + list(sage.parameters()))
        xr = x.clone().requires_grad_(True)
        ref2 = torch.relu((torch.zeros(nd, 10).index_add_(0, blk[1], xr[blk[0]]) / cnt) @ sage.fc_neigh.weight.t()
                          + xr[:nd] @ sage.fc_self.weight.t() + sage.bias)
        gref = torch.autograd.grad(ref2.square().sum(), [xr] + list(sage.parameters()))
        for a, b in zip(gr, gref):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    layers.FUSED_MIN_EDGES = 2_000_000
    # a float64 edge_weight promotes the messages exactly as the reference's message() route does
    conv = layers.GCNConv(10, 8, norm="none")
    ew64 = torch.rand(ei.shape[1], generator=g, dtype=torch.float64)
    y64 = conv(x, ei, ew64)
    ref64 = agg((x @ conv.linear.weight.t()).double(), ew64) if False else \
        torch.zeros(N, 8, dtype=torch.float64).index_add_(0, dst, (x @ conv.linear.weight.t())[src] * ew64.unsqueeze(1)) + conv.bias
    assert y64.dtype == torch.float64
    torch.testing.assert_close(y64, ref64, rtol=1e-9, atol=1e-9)
    pool = layers.SAGEConv(10, 6, aggr="pool")
    hp = torch.relu(pool.pool(x))
    mx = torch.full((nd, 10), -3.4028234663852886e38).scatter_reduce(0, blk[1].view(-1, 1).expand(-1, 10), hp[blk[0]], "amax")
    torch.testing.assert_close(pool((x, x[:nd]), blk), pool.fc_neigh(mx) + pool.fc_self(x[:nd]) + pool.bias,
                               rtol=1e-5, atol=1e-5)
    # aggr='lstm' (sage_conv.py:93-98): the source rows as [N_dst, fan-out, D] sequences, last hidden state
    lst = layers.SAGEConv(10, 6, aggr="lstm")
    xs = torch.randn(nd * 3, 10, generator=g)
    want = lst.fc_neigh(lst.lstm(xs.reshape(nd, 3, 10))[1][0][0]) + lst.fc_self(x[:nd]) + lst.bias
    torch.testing.assert_close(lst((xs, x[:nd]), blk), want, rtol=1e-6, atol=1e-6)
    # GATConv == FusedGATConv, narrow and 41 channels per head (padded inside the fused layer)
    for (C, concat) in ((8, True), (41, False)):
        gat = layers.GATConv(10, C, heads=4, concat=concat)
        fgat = layers.FusedGATConv(10, C, heads=4, concat=concat)
        fgat.load_state_dict(gat.state_dict())
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = gat(xa, ei, N), fgat(xb, ei, N)
        torch.testing.assert_close(ya, yb, rtol=1e-5, atol=1e-6)
        ya.square().sum().backward()
        yb.square().sum().backward()
        torch.testing.assert_close(xa.grad, xb.grad, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(gat.w.grad, fgat.w.grad, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(gat.att.grad, fgat.att.grad, rtol=2e-4, atol=2e-5)
        # the prebuilt CSR / CSC / permute keyword arguments (fusedgat_conv.py:95-117 builds exactly these), int32
        from gammagl_amd import sparse
        s1 = sparse.sort_edge_index(torch.stack([ei[1], ei[0]]), num_nodes=N, eng=eng)       # rows = aggregating nodes
        s2, permute = sparse.sort_edge_index(s1, torch.arange(ei.shape[1]), N, sort_by_row=False, eng=eng)
        kw = dict(row_ptr=sparse.ind2ptr(s1[0], N, eng=eng).int(), col_ind=s1[1].int(),
                  col_ptr=sparse.ind2ptr(s2[1], N, eng=eng).int(), row_ind=s2[0].int(), permute=permute.int())
        xc = x.clone().requires_grad_(True)
        built = eng.stats["plans_built"]
        yc = fgat(xc, None, N, **kw)
        torch.testing.assert_close(yc, yb, rtol=1e-5, atol=1e-6)
        fgat.zero_grad()
        yc.square().sum().backward()
        torch.testing.assert_close(xc.grad, xb.grad, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(gat.att.grad, fgat.att.grad, rtol=2e-4, atol=2e-5)
        fgat(x, None, N, **kw)
        assert eng.stats["plans_built"] == built + 2      # CSR + CSC taken as given, once; the second call hits the cache
    # models: GAT (fused, dropout on) trains a step; GraphSAGE sample model + layer-wise inference; full model
    gm = layers.GATModel(10, 8, 5, heads=4, drop_rate=0.5, num_layers=2, fused=True)
    gm.train()
    gm(x, ei, N).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in gm.parameters())
    ns = NeighborSampler(ei, [5, 3], num_nodes=N, eng=eng)
    sm = layers.GraphSAGESampleModel(10, 8, 5, drop_rate=0.0, num_layers=2)
    dst_n, n_id, adjs = ns.sample(torch.arange(6))
    assert sm(x[n_id], adjs).shape == (6, 5)
    full = NeighborSampler(ei, [-1], num_nodes=N, eng=eng)
    sm.eval()
    logits = sm.inference(x, full, batch_size=16)
    fm_ref = x
    for i, layer in enumerate(sm.convs):  # full-neighbourhood layer-wise == the model applied to the whole graph
        fm_ref = layer(fm_ref, ei)
    torch.testing.assert_close(logits, fm_ref, rtol=1e-5, atol=1e-5)
    # the static-shape block trainer (graph-capturable step) == the dynamic sampler's trainer when every
    # neighbourhood is kept whole (fan-out >= the largest degree): same losses, same weights after two steps
    from gammagl_amd.sampler import BlockSampler
    from gammagl_amd.trainer import SAGEBlockTrainer, SAGETrainer
    eu = torch.unique(torch.randint(0, 120, (2, 700), generator=g), dim=1)
    degu = torch.bincount(eu[1], minlength=120)
    xu, yu = torch.randn(120, 10, generator=g), torch.randint(0, 4, (120,), generator=g)
    fan = int(degu.max())
    for hid in (8, 16):   # 16: the first layer's input (10) is narrower than its output -> the block path aggregates first
        tb = SAGEBlockTrainer(BlockSampler(eu, [fan, fan], num_nodes=120, eng=eng), 10, hid, 4, seed=3, device="cpu")
        td = SAGETrainer(NeighborSampler(eu, [-1, -1], num_nodes=120, eng=eng), 10, hid, 4, seed=3, device="cpu")
        td.opt = torch.optim.Adam(td.net.parameters(), lr=0.005)
        for it in range(2):
            sd = torch.randperm(120, generator=g)[:8]
            lb, ld = tb.step(xu, yu, sd), td.step(xu, yu, sd)
            torch.testing.assert_close(lb, ld, rtol=1e-5, atol=1e-6)
        for pb, pd in zip(tb.net.parameters(), td.net.parameters()):
            torch.testing.assert_close(pb.detach(), pd.detach(), rtol=2e-4, atol=2e-6)
    fm = layers.GraphSAGEFullModel(10, 8, 5, 1, torch.relu, 0.0, "mean")
    assert fm(x, ei).shape == (N, 5)
    print("LAYERS_OK")


def test_layer_classes_on_the_emulated_engine():
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    p = subprocess.run([sys.executable, __file__, "body"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "LAYERS_OK" in p.stdout, (p.stdout + p.stderr)[-3000:]


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "body":
    _body()
