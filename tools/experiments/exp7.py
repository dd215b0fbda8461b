"""Does a burst of bf16 matrix work slow down the recurrent kernel that follows it?  (last paragraph)
One cfg2-shaped layer (B = 32, T = 500, D = 2048, H = 512), persistent recurrence forward + backward, preceded by
EXP_BURST = none | pk (six-plane bf16 product, ~2 ms) | f32 (exact-fp32 product, ~2 ms) | copy (2 ms of HBM streaming) |
idle (2 ms of host sleep after a sync).  Prints us per sequential step (events around the recurrent launches)."""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from nabu_amd import ops  # noqa: E402

B, T, D, H = 32, int(os.environ.get('EXP_T', '500')), 2048, 512
x = torch.randn(B, T, D, device='cuda') * 0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda') * 0.03 for s in [(D + H, 4 * H), (4 * H,), (D + H, 4 * H), (4 * H,)]]
dout = torch.randn(B, T, 2 * H, device='cuda')
plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_PERSISTENT, 'f32')
out = torch.zeros(B, T, 2 * H, device='cuda')
reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = [torch.zeros_like(q) for q in p]
dx = torch.zeros_like(x)
A = torch.randn(16000, 8192, device='cuda')
Bm = torch.randn(2048, 8192, device='cuda')
C = torch.empty(16000, 2048, device='cuda')
pa, pb = ops.PackedOperand(16000, 8192, 3, 'cuda'), ops.PackedOperand(2048, 8192, 3, 'cuda')
ops.pk_pack(pa, A)
ops.pk_pack(pb, Bm)
big = torch.randn(256 << 20, device='cuda')


def burst(kind):
    if kind.startswith('pk'):
        for _ in range(int(kind[2:] or 1)):
            ops.gemm_pk(pa, pb, C, 3)
    elif kind == 'f32':
        ops.gemm(A[:8000], Bm, C[:8000], False, True, precision='f32')
    elif kind == 'copy':
        for _ in range(4):
            big.mul_(1.0000001)
    elif kind == 'idle':
        torch.cuda.synchronize()
        time.sleep(0.002)


prof = ops.enable_profiler()
for kind in os.environ.get('EXP_BURST', 'none,pk,pk2,pk3,f32,copy,idle,none').split(','):
    for it in range(8):
        burst(kind)
        ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
        burst(kind)
        ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0], g[1], g[2], g[3])
    torch.cuda.synchronize()
    recs = prof.collect()
    fw = [r[4] * 1e3 / T for r in recs if r[0] == 'fwd'][2:]
    bw = [r[4] * 1e3 / T for r in recs if r[0] != 'fwd'][2:]
    print('burst %-5s fwd %.3f us/step  bwd %.3f us/step' % (kind, sum(fw) / len(fw), sum(bw) / len(bw)), flush=True)
