import sys; sys.path.insert(0,'.')
import torch, numpy as np
from nabu_amd import ops
def t(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K1, K2) in [(32, 2048, 1024, 512), (32, 512, 512, 0), (32, 1024, 2048, 0), (32, 512, 2048, 0), (64, 2048, 1024, 512)]:
    a = torch.randn(M, K1, device='cuda'); b = torch.randn(K1, N, device='cuda')
    a2 = torch.randn(M, K2, device='cuda') if K2 else None; b2 = torch.randn(K2, N, device='cuda') if K2 else None
    c = torch.zeros(M, N, device='cuda'); c2 = torch.zeros(M, N, device='cuda')
    def old():
        ops.gemm(a, b, c)
        if K2: ops.gemm(a2, b2, c, beta=1.0)
    def new():
        ops.gemm2(a, b, a2, b2, c2)
    to, tn = t(old), t(new)
    old(); new(); torch.cuda.synchronize()
    err = (c - c2).abs().max().item() / c.abs().max().item()
    print('M=%d N=%d K=%d+%d: old %.1f us, fused (incl. 1 memset) %.1f us, rel diff %.1e' % (M, N, K1, K2, to, tn, err))
