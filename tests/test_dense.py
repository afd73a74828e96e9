"""gammagl_amd.dense: the split-reduction weight gradient equals g^T @ x and Linear keeps nn.Linear's
parameters, outputs and gradients."""
import torch

from gammagl_amd import dense


def test_wgrad_split_matches_single_gemm(monkeypatch):
    monkeypatch.setattr(dense, "ROWS_PER_SPLIT", 16)
    g = torch.Generator().manual_seed(0)
    for n in (5, 31, 32, 1003, 16 * 512 * 2 + 7):   # below the split threshold, ragged tail, MAX_SPLITS cap
        a = torch.randn(n, 7, generator=g, dtype=torch.float64)
        b = torch.randn(n, 5, generator=g, dtype=torch.float64)
        torch.testing.assert_close(dense.wgrad(a, b), a.t() @ b, rtol=1e-12, atol=1e-11)


def test_linear_is_nn_linear(monkeypatch):
    monkeypatch.setattr(dense, "ROWS_PER_SPLIT", 8)
    torch.manual_seed(0)
    for bias in (False, True):
        lin = dense.Linear(6, 4, bias=bias).double()
        ref = torch.nn.Linear(6, 4, bias=bias).double()
        ref.load_state_dict(lin.state_dict())
        xa = torch.randn(100, 6, dtype=torch.float64, requires_grad=True)
        xb = xa.detach().clone().requires_grad_()
        ya, yb = lin(xa), ref(xb)
        torch.testing.assert_close(ya, yb)
        ya.square().sum().backward()
        yb.square().sum().backward()
        torch.testing.assert_close(xa.grad, xb.grad)
        torch.testing.assert_close(lin.weight.grad, ref.weight.grad, rtol=1e-12, atol=1e-11)
        if bias:
            torch.testing.assert_close(lin.bias.grad, ref.bias.grad)
    assert lin(torch.randn(2, 3, 6, dtype=torch.float64)).shape == (2, 3, 4)  # >2-D input: nn.Linear path
