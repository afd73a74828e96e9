"""The dense step either side of the aggregate: y = x @ W^T on `[N, in]` node features (gcn_conv.py:79,
sage_conv.py:100-108).  The products themselves are library GEMMs (hipBLASLt through torch); what is
ours is the shape handling of the WEIGHT gradient gW[out,in] = g^T[out,N] @ x[N,in]: N is millions,
out/in are a few hundred, so a single GEMM has only (out/32)*(in/32) output tiles for 256 CUs and one
very long reduction per tile (hipBLASLt picks MT32x32x256: 5.1 ms at N = 2.45 M, 256x256).  Splitting
the N axis into S batched GEMMs and summing the S partial products fills the chip: 2.2 ms for the same
product, 0.67 vs 3.9 ms for 47x256 (tools/wgrad_probe.py, profiles/r1_wgrad_probe.txt).
"""
import torch

ROWS_PER_SPLIT = 8192
MAX_SPLITS = 512


def wgrad(g, x):
    """g^T @ x for row-major g [N, out], x [N, in] with the N axis split (see module docstring)."""
    n = g.shape[0]
    S = min(MAX_SPLITS, n // ROWS_PER_SPLIT)
    if S < 2 or g.dim() != 2 or x.dim() != 2:
        return g.t() @ x
    m = (n // S) * S
    gw = torch.bmm(g[:m].view(S, m // S, -1).transpose(1, 2), x[:m].view(S, m // S, -1)).sum(0)
    if m < n:
        gw.addmm_(g[m:].t(), x[m:])
    return gw


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = wgrad(g.contiguous(), x) if ctx.needs_input_grad[1] else None
        return gx, gw


class _MatmulFn(torch.autograd.Function):
    """x @ w for node features x [N, in] and a weight w [in, out] (GATConv's `tlx.matmul(x, self.w)`, gat_conv.py:99): the weight
    gradient x^T g is a reduction over N = 10^5 .. 10^6 nodes — taken through `wgrad` (the node axis split into slabs, partial
    products summed: a two-level sum) instead of ONE GEMM with an N-long accumulation per output element.  Round 6: with
    bench.py's tuned GEMM selection the single-GEMM form picked kernels whose sequential f32 accumulation over 233 k terms left the
    GAT model's weight gradients 2.5e-4 .. 1.2e-3 from a float64 evaluation (the default heuristics: 1.5e-5); the slab form does
    not depend on the selection."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w.t() if ctx.needs_input_grad[0] else None
        gw = wgrad(x, g.contiguous()) if ctx.needs_input_grad[1] else None
        return gx, gw


def node_matmul(x, w):
    """x @ w with the weight gradient as a slab-split sum (see _MatmulFn) where x is a contiguous [N, in] float matrix."""
    if x.dim() == 2 and x.is_contiguous() and w.dim() == 2 and x.is_floating_point() and x.dtype == w.dtype:
        return _MatmulFn.apply(x, w)
    return x @ w


class Linear(torch.nn.Linear):
    """torch.nn.Linear (same parameters / state_dict) whose weight gradient uses `wgrad` for 2-D inputs."""

    def forward(self, x):
        if x.dim() != 2 or not x.is_contiguous():
            return super().forward(x)
        y = _LinearFn.apply(x, self.weight)
        return y if self.bias is None else y + self.bias
