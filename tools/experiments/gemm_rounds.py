import sys; sys.path.insert(0,'.')
import torch
from nabu_amd import ops
def bench(ta, tb, M, N, K, reps=8):
    a = torch.randn((K, M) if ta else (M, K), device='cuda')
    b = torch.randn((N, K) if tb else (K, N), device='cuda')
    c = torch.empty(M, N, device='cuda')
    ops.gemm(a, b, c, ta, tb); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, ta, tb)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    tiles = ((M + 127)//128) * ((N + 127)//128)
    print('%s%s M=%6d N=%5d K=%6d tiles %5d (%.2f rounds of 768): %.3f ms  %6.1f TF/s' % ('T' if ta else 'N', 'T' if tb else 'N', M, N, K, tiles, tiles/768, ms, 2*M*N*K/ms/1e9))
for M in (3072, 6144, 8000, 9216, 12288, 16000, 18432, 24576, 32000):
    bench(0, 0, M, 2048, 2048)
for M in (6144, 12288, 16000, 18432):
    bench(0, 1, M, 2048, 2048)
for K in (1024, 2048, 4096):
    bench(0, 0, 12288, 2048, K)
