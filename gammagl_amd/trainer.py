"""Training-step harness: what examples/gcn/gcn_trainer.py:51-117 does per
epoch — SemiSpvzLoss (forward, gather train rows, softmax cross-entropy) + TrainOneStep (backward,
Adam with weight decay) — in plain torch around the gammagl_amd layers."""
import os

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
        if logits.shape[0] != seeds.shape[0]:     # (a full-range slice still records a SliceBackward: a zero-fill + a copy per step)
            logits = logits[: seeds.shape[0]]
        loss = F.cross_entropy(logits, y.index_select(0, seeds))
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

    # ---- replicas (world > 1): the step as TWO hipGraphs around the one collective ---------------------------------
    def _grads_to_flat(self):
        o = 0
        for p in self._params:
            n = p.numel()
            self._flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n

    def _flat_to_grads(self):
        self._flat.div_(self.world)
        o = 0
        for p in self._params:
            n = p.numel()
            p.grad.copy_(self._flat[o:o + n].view_as(p.grad))
            o += n

    def _front(self, x, y, seeds):
        """sample + gather + forward + loss + backward + the gradients flattened into one persistent buffer"""
        self.net.train()
        self.opt.zero_grad(set_to_none=True)
        n_id, blocks, _ = self.sampler.sample(seeds, caps=self.caps)
        logits = self.net(x.index_select(0, n_id), blocks)
        if logits.shape[0] != seeds.shape[0]:     # (a full-range slice still records a SliceBackward: a zero-fill + a copy per step)
            logits = logits[: seeds.shape[0]]
        loss = F.cross_entropy(logits, y.index_select(0, seeds))
        loss.backward()
        self._grads_to_flat()
        return loss.detach()

    def _back(self):
        """the averaged gradients back into place + Adam"""
        self._flat_to_grads()
        self.opt.step()

    def capture(self, x, y, seeds, warmup=3):
        """Record step(x, y, seeds) into a hipGraph.  Afterwards: seeds.copy_(new_batch); trainer.replay().

        world > 1 (replicas): RCCL collectives do not record on this stack (profiles/r3_rccl_capture_attempt.txt, re-tried
        in round 4: profiles/r4_rccl_capture_retry.txt), so the step becomes TWO graphs with the one collective between
        them — [sample, gather, forward, loss, backward, flatten gradients] | all-reduce (eager, ~300 KB) | [unflatten,
        Adam] — instead of ~190 eager launches per batch; both graphs share one memory pool (the second reads the
        gradient tensors the first one's backward allocated)."""
        if self.world == 1:
            self.graph = GraphedStep(lambda: self.step(x, y, seeds), warmup=warmup)
            return self.graph
        import torch.distributed as dist

        # the FIRST warm-up step runs through the eager step(): which parameters receive a gradient at all is only known
        # after a backward — the flat buffer covers exactly those (step() filters `p.grad is not None` the same way; a
        # parameter without a gradient used to crash _grads_to_flat)
        self.step(x, y, seeds)
        self._params = [p for p in self.net.parameters() if p.grad is not None]
        self._flat = torch.zeros(sum(p.numel() for p in self._params), device=x.device, dtype=torch.float32)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1) - 1):  # plans, sampler scratch, Adam state, RCCL channels exist before capture
                self._front(x, y, seeds)
                dist.all_reduce(self._flat, group=self.group)
                self._back()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if os.environ.get("GGL_SAGE_ONE_GRAPH", "0") == "1":
            # opt-in: the collective INSIDE the graph.  With a pre-warmed communicator (the warm-up steps above) an RCCL
            # all-reduce does record and replay on this stack (profiles/r4_rccl_capture_retry.txt, tools/rccl_capture_retry.py);
            # round 3's crash was the halo step's uneven all-to-all-v.  Not the default: verified on a one-rank group only.
            self._g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g1, capture_error_mode="thread_local"):
                self._loss = self._front(x, y, seeds)
                dist.all_reduce(self._flat, group=self.group)
                self._back()
            self.graph = lambda: (self._g1.replay(), self._loss)[1]
            return self.graph
        pool = torch.cuda.graph_pool_handle()
        self._g1, self._g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # "thread_local": an RCCL group is live — its watchdog thread polls events, which the default "global" capture
        # mode turns into a capture error (GraphedStep's docstring; the ONE_GRAPH branch above does the same)
        with torch.cuda.graph(self._g1, pool=pool, capture_error_mode="thread_local"):
            self._loss = self._front(x, y, seeds)
        with torch.cuda.graph(self._g2, pool=pool, capture_error_mode="thread_local"):
            self._back()
        self.graph = self._replay_replica
        return self.graph

    def _replay_replica(self):
        import torch.distributed as dist

        self._g1.replay()
        dist.all_reduce(self._flat, group=self.group)
        self._g2.replay()
        return self._loss

    def replay(self):
        return self.graph()
