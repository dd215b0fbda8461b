"""How long do the first launches of the persistent recurrent kernels take in a fresh process?  (the r04/r05 cfg1 kernel
statistics show ONE forward launch of 16-35 ms among 0.2 ms ones)  Times the first calls of a cfg1-shaped and of a
cfg2-shaped layer with events, optionally after a warm-up of plain kernels."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nabu_amd import ops


def layer(B, T, D, H, n=4, tag=''):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(B, T, D, device='cuda', generator=g)
    p = [torch.randn(D + H, 4 * H, device='cuda', generator=g) * 0.1, torch.zeros(4 * H, device='cuda'),
         torch.randn(D + H, 4 * H, device='cuda', generator=g) * 0.1, torch.zeros(4 * H, device='cuda')]
    lens = torch.full((B,), T, dtype=torch.int32, device='cuda')
    plan = ops.BlstmPlan(B, T, D, H, T)
    out = torch.empty(B, T, 2 * H, device='cuda')
    reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ops.check_persist_status()
    print('%s B %d T %d D %d H %d: forward calls (ms): %s' % (tag, B, T, D, H, ' '.join('%.3f' % t for t in ts)), flush=True)


if __name__ == '__main__':
    order = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
    if order == 'warm':
        a = torch.randn(4096, 4096, device='cuda')
        for _ in range(20):
            a = a * 1.0001 + 0.1
        torch.cuda.synchronize()
        order = 'cfg1'
    if order == 'cfg1':
        layer(8, 200, 40, 256, tag='first')
        layer(8, 200, 512, 256, tag='second')
        layer(32, 1000, 40, 512, tag='third')
    else:
        layer(32, 1000, 40, 512, tag='first')
        layer(8, 200, 40, 256, tag='second')
