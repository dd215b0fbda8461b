"""Phase stamps of block 0 / wave 0 inside ONE backward step of lstm_mxh_bwd_kernel (NABU_PERSIST_DEBUG=4; 10 ns ticks):
usage: NABU_PERSIST_DEBUG=4 python tools/experiments/bwd_stamps.py   (add 1 to the debug value: no waiting)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from nabu_amd import ops, _hip
B, T, D, H = 32, 500, 2048, 512
x = torch.randn(B, T, D, device='cuda') * 0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda') * 0.03 for s in [(D + H, 4 * H), (4 * H,), (D + H, 4 * H), (4 * H,)]]
dout = torch.randn(B, T, 2 * H, device='cuda')
plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_PERSISTENT)
out = torch.zeros(B, T, 2 * H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = [torch.zeros_like(q) for q in p]; dx = torch.zeros_like(x)
prof = ops.enable_profiler()
acc = []
for it in range(8):
    ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
    ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0], g[1], g[2], g[3])
    torch.cuda.synchronize()
    ws = _hip.Workspace._bufs[(str(x.device), 'blstm')]
    st = ws[4 * (320 + 32):4 * (320 + 64)].view(torch.int32).cpu().numpy().astype(np.int64)
    if it >= 2:
        acc.append(st.copy())
recs = prof.collect()
print('dbg', os.environ.get('NABU_PERSIST_DEBUG', '0'), 'bwd us/step', ['%.2f' % (r[4] * 1e3 / T) for r in recs if r[0] == 'bwd'][2:])
order = [(0, 'step start'), (1, 'poll done'), (2, 'dz planes written'), (3, 'barrier passed'), (10, 'first matrix group issued'),
         (11, 'half 0 matrix done'), (4, 'half 0 folded'), (14, 'half 0 published'), (12, 'half 1 matrix done'), (13, 'half 1 folded'),
         (9, 'half 1 published'), (5, 'step end')]
m = np.array(acc)
prev = None
for i, name in order:
    d = ((m[:, i] - m[:, 0]) & 0xffffffff) * 10
    print('%-28s +%5.0f ns (median over %d launches; min %d max %d)%s' % (name, np.median(d), len(d), d.min(), d.max(),
          '' if prev is None else '   delta %4.0f' % (np.median(d) - prev)))
    prev = np.median(d)
