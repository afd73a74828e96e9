"""Training-step harness: what examples/gcn/gcn_trainer.py:51-117 does per
epoch — SemiSpvzLoss (forward, gather train rows, softmax cross-entropy) + TrainOneStep (backward,
Adam with weight decay) — in plain torch around the gammagl_amd layers."""
import torch
import torch.nn.functional as F

from .layers import GCNModel, GraphSAGESampleModel


class GCNTrainer:
    def __init__(self, feature_dim, hidden_dim, num_class, num_layers=3, drop_rate=0.5, lr=0.01,
                 l2_coef=5e-4, norm="both", seed=0, device="cuda"):
        torch.manual_seed(seed)
        self.net = GCNModel(feature_dim, hidden_dim, num_class, drop_rate=drop_rate,
                            num_layers=num_layers, norm=norm).to(device)
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, weight_decay=l2_coef)

    def loss(self, x, edge_index, y, train_idx, num_nodes):
        logits = self.net(x, edge_index, None, num_nodes)
        return F.cross_entropy(logits[train_idx], y[train_idx])

    def step(self, x, edge_index, y, train_idx, num_nodes):
        self.net.train()
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(x, edge_index, y, train_idx, num_nodes)
        loss.backward()
        self.opt.step()
        return loss.detach()


class GraphedStep:
    """Capture a whole training step in a hipGraph and replay it.

    For Cora-sized graphs a step is ~100 tiny launches and the GPU idles between them; the C ABI never
    syncs or allocates on the hot path, so the step — our kernels, hipBLASLt, the side-stream weight
    gradients, Adam(capturable=True) — captures as is (measured 1.15 -> 0.50 ms/step on a Cora-sized
    graph, no change on an arxiv-sized one, which is GPU-bound).  Plans must exist before capture, so a
    few eager warm-up steps run first; the captured step reads its inputs from the tensors it was
    captured with (update them in place)."""

    def __init__(self, step_fn, warmup=3, capture_error_mode="global"):
        """`capture_error_mode="thread_local"`: for steps that contain RCCL collectives — the process group's watchdog
        thread polls events while the capture is open, which the default global mode treats as a capture violation."""
        self.step_fn = step_fn
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.out = step_fn()

    def __call__(self):
        self.graph.replay()
        return self.out


class SAGETrainer:
    """Neighbour-sampled GraphSAGE training (examples/graphsage/reddit_sage_trainer.py:45-105): per batch,
    sample the blocks on the device (NeighborSampler), gather the input rows, SAGEConv(mean) per hop,
    softmax cross-entropy on the seed nodes, Adam.

    Multi-GPU (BASELINE config 4): the aggregate does not shard — every rank is a replica
    that samples its OWN seeds from the replicated graph; the only exchange is one flat all-reduce of the
    weight gradients per step (`group` = a torch.distributed process group; RCCL on the GPUs)."""

    def __init__(self, sampler, in_feat, hid_feat, num_class, num_layers=2, drop_rate=0.0, lr=0.005, seed=0,
                 device="cuda", group=None, world=1):
        self.sampler, self.group, self.world = sampler, group, int(world)
        torch.manual_seed(seed)  # identical initial weights on every replica
        self.net = GraphSAGESampleModel(in_feat, hid_feat, num_class, drop_rate, num_layers).to(device)
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr)

    def step(self, x, y, seeds):
        """One optimizer step on this rank's `seeds`; the gradient is the mean over ALL ranks' seeds when
        every rank passes the same number of seeds."""
        import torch.distributed as dist

        self.net.train()
        self.opt.zero_grad(set_to_none=True)
        dst, n_id, adjs = self.sampler.sample(seeds)
        logits = self.net(x[n_id], adjs)
        loss = F.cross_entropy(logits, y[dst])
        loss.backward()
        if self.world > 1:
            params = [p for p in self.net.parameters() if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat, group=self.group)
            flat.div_(self.world)
            o = 0
            for p in params:
                n = p.grad.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        self.opt.step()
        return loss.detach()


class SAGEBlockTrainer:
    """The same mini-batch GraphSAGE step on the static-shape BlockSampler: sampling, the feature gather, the
    SAGEConv layers (aggregate + "+ fc_self + bias -> ReLU" in one kernel each), the loss on the seed rows,
    backward (block CSC built on the device) and Adam run without a single host read — `capture()` records the
    whole step into one hipGraph that is replayed per batch with the seeds updated in place."""

    def __init__(self, sampler, in_feat, hid_feat, num_class, num_layers=2, drop_rate=0.0, lr=0.005, seed=0,
                 device="cuda", capturable=None, caps=None, group=None, world=1):
        """`caps`: block capacities from `sampler.calibrate(batch_size)` (default: the worst case, on which
        every dense layer runs several times more padding than data); check `sampler.overflow_count()`.
        `world` > 1: replicas (BASELINE config 4's multi-GPU mode) — every rank samples its own seeds and the
        weight gradients are averaged by one flat all-reduce per step; that step runs eagerly (the collective is
        not recorded into the graph), `capture()` is for world == 1."""
        self.sampler, self.caps, self.group, self.world = sampler, caps, group, int(world)
        torch.manual_seed(seed)
        self.net = GraphSAGESampleModel(in_feat, hid_feat, num_class, drop_rate, num_layers).to(device)
        cap = torch.device(device).type == "cuda" if capturable is None else capturable
        # one fused optimizer kernel and gradients assigned (not zero-filled and accumulated) per step: in a replayed
        # graph the step is bound by the NUMBER of small kernels (~5 us + gap each), not by their work
        try:
            self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, capturable=cap, fused=cap)
        except (RuntimeError, TypeError):
            self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, capturable=cap)
        self.graph = None

    def step(self, x, y, seeds):
        self.net.train()
        self.opt.zero_grad(set_to_none=True)
        n_id, blocks, _ = self.sampler.sample(seeds, caps=self.caps)
        logits = self.net(x.index_select(0, n_id), blocks)
        loss = F.cross_entropy(logits[: seeds.shape[0]], y.index_select(0, seeds))
        loss.backward()
        if self.world > 1:
            import torch.distributed as dist

            params = [p for p in self.net.parameters() if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat, group=self.group)
            flat.div_(self.world)
            o = 0
            for p in params:
                n = p.grad.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        self.opt.step()
        return loss.detach()

    def capture(self, x, y, seeds, warmup=3):
        """Record step(x, y, seeds) into a hipGraph.  Afterwards: seeds.copy_(new_batch); trainer.replay()."""
        if self.world > 1:
            raise RuntimeError("capture() records a single-replica step; with world > 1 call step() per batch")
        self.graph = GraphedStep(lambda: self.step(x, y, seeds), warmup=warmup)
        return self.graph

    def replay(self):
        return self.graph()
