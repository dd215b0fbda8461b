import sys; sys.path.insert(0,'.')
import torch
from nabu_amd import ops
def bench(ta, tb, M, N, K, reps=10):
    a = torch.randn((K, M) if ta else (M, K), device='cuda')
    b = torch.randn((N, K) if tb else (K, N), device='cuda')
    c = torch.empty(M, N, device='cuda')
    for _ in range(3): ops.gemm(a, b, c, ta, tb)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, ta, tb)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print('%s%s M=%6d N=%5d K=%6d: %.3f ms  %6.1f TF/s' % ('T' if ta else 'N', 'T' if tb else 'N', M, N, K, ms, 2*M*N*K/ms/1e9))
bench(0, 0, 18432, 2048, 2048); bench(0, 1, 18432, 2048, 2048); bench(1, 0, 2048, 2048, 18432)
