"""What does a recurrent kernel lose when memory-bound operand packs run BESIDE it on a second stream?  (round 5: the
packs and maxima a layer's weight-gradient products need do not depend on the backward pass — could they hide under
the recurrences?)  One cfg2-shaped layer (B = 32, T = 500, D = 2048, H = 512), persistent recurrence forward + backward on
the main stream; on a side stream, EXP_SIDE = none | packT (transposed f16x3 packs of a [16000, 2048] matrix, the x^T /
h^T kind, ~30 us each) | packN (natural packs) | copy (elementwise streaming) in a loop for the duration.
Prints us per sequential step and how many side kernels completed per millisecond of recurrence."""
import os
import sys

import torch

sys.path.insert(0, '.')
from nabu_amd import ops  # noqa: E402

B, T, D, H = 32, int(os.environ.get('EXP_T', '500')), int(os.environ.get('EXP_D', '2048')), 512
x = torch.randn(B, T, D, device='cuda') * 0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda') * 0.03 for s in [(D + H, 4 * H), (4 * H,), (D + H, 4 * H), (4 * H,)]]
dout = torch.randn(B, T, 2 * H, device='cuda')
plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_PERSISTENT, 'f32')
out = torch.zeros(B, T, 2 * H, device='cuda')
reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = [torch.zeros_like(q) for q in p]
dx = torch.zeros_like(x) if D >= 256 else None
src = torch.randn(16000, 2048, device='cuda')
poT = ops.PackedOperand(2048, 16000, 2, 'cuda')
poN = ops.PackedOperand(16000, 2048, 2, 'cuda')
big = torch.randn(64 << 20, device='cuda')
side = torch.cuda.Stream()
NSIDE = int(os.environ.get('EXP_NSIDE', '24'))


def side_work(kind):
    n = 0
    with torch.cuda.stream(side):
        for _ in range(NSIDE):
            if kind == 'packT':
                ops.pk_pack(poT, src, transposed=True, bound=1.0)
            elif kind == 'packN':
                ops.pk_pack(poN, src, bound=1.0)
            elif kind == 'copy':
                big.mul_(1.0000001)
            n += 1
    return n


prof = ops.enable_profiler()
for kind in os.environ.get('EXP_SIDE', 'none,packT,packN,copy,none').split(','):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot_side = 0.0
    for it in range(8):
        torch.cuda.synchronize()
        if kind != 'none':
            with torch.cuda.stream(side):
                e0.record()
            side_work(kind)
            with torch.cuda.stream(side):
                e1.record()
        ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
        torch.cuda.synchronize()
        if kind != 'none':
            tot_side += e0.elapsed_time(e1)
            side_work(kind)
        ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0], g[1], g[2], g[3])
    torch.cuda.synchronize()
    ops.check_persist_status()
    recs = prof.collect()
    fw = [r[4] * 1e3 / T for r in recs if r[0] == 'fwd'][2:]
    bw = [r[4] * 1e3 / T for r in recs if r[0] != 'fwd'][2:]
    print('side %-6s fwd %.3f us/step  bwd %.3f us/step   (side batch of %d: %.3f ms beside the forward kernel)'
          % (kind, sum(fw) / len(fw), sum(bw) / len(bw), NSIDE, tot_side / 8), flush=True)
