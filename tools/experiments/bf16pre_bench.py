import sys; sys.path.insert(0,'.')
import torch
from nabu_amd import ops
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (M, N, K) in [(51200, 2048, 2048), (25600, 2048, 2048), (12800, 2048, 2048), (2048, 2048, 51200), (2048, 2048, 12800)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); c = torch.empty(M, N, device='cuda')
    ab, bb = ops.cvt_bf16(a), ops.cvt_bf16(b)
    bt = b.t().contiguous()
    t_old = t(lambda: ops.gemm(a, bt, c, precision='bf16'))
    t_new = t(lambda: ops.gemm_bf16_nt(ab, bb, c))
    t_cvt = t(lambda: ops.cvt_bf16(a)); t_cvtT = t(lambda: ops.cvt_bf16(a, transpose=True))
    fl = 2.0 * M * N * K
    print('M=%6d N=%5d K=%6d: in-kernel rounding %.3f ms (%.0f TF/s) | resident bf16 %.3f ms (%.0f TF/s) | cvt A %.3f ms, cvt A^T %.3f ms' % (
        M, N, K, t_old, fl / t_old / 1e9, t_new, fl / t_new / 1e9, t_cvt, t_cvtT))
