import torch, sys
sys.path.insert(0, '/root/repo')
from gammagl_amd.dense import wgrad
dev = torch.device('cuda')
n = 2449029
def ev(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for k in (100, 104, 112, 128):
    x = torch.randn(n, k, device=dev); w = torch.randn(256, k, device=dev); g = torch.randn(n, 256, device=dev)
    print(k, 'fwd x@w.T %.2f ms' % ev(lambda: x @ w.t()), ' wgrad %.2f ms' % ev(lambda: wgrad(g, x)), ' plain g.T@x %.2f ms' % ev(lambda: g.t() @ x))
# strided input: x128 view of first 100 cols? (lda=128, K=100)
x128 = torch.randn(n, 128, device=dev); w = torch.randn(256, 100, device=dev)
print('strided K=100 lda=128 fwd %.2f ms' % ev(lambda: x128[:, :100] @ w.t()))
