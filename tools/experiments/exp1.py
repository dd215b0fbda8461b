import sys, os, time; sys.path.insert(0,'.')
import numpy as np, torch
from nabu_amd import ops, _hip
import os as _os
if _os.environ.get('NABU_LIB'): _hip.LIB_PATH = _os.path.abspath(_os.environ['NABU_LIB'])
B,T,D,H = int(_os.environ.get("EXP_B","32")),int(_os.environ.get("EXP_T","500")),2048,512
x = torch.randn(B,T,D, device='cuda')*0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda')*0.03 for s in [(D+H,4*H),(4*H,),(D+H,4*H),(4*H,)]]
dout = torch.randn(B,T,2*H, device='cuda')
plan = ops.BlstmPlan(B,T,D,H,T,ops.LSTM_PERSISTENT)
out = torch.zeros(B,T,2*H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = [torch.zeros_like(q) for q in p]; dx = torch.zeros_like(x)
prof = ops.enable_profiler()
for it in range(6):
    ops.blstm_fwd(plan, x, lens, p[0],p[1],p[2],p[3], out, reserve)
    ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0],g[1],g[2],g[3])
torch.cuda.synchronize()
recs = prof.collect()
print('dbg', os.environ.get('NABU_PERSIST_DEBUG','0'), ' '.join('%s %.2f us/step' % (r[0], r[4]*1e3/T) for r in recs[4:]))
ws = _hip.Workspace._bufs[(str(x.device),'blstm')]
xcc = ws[64:64+4*256].view(torch.int32).cpu().numpy()
print('status', int(ws[:4].view(torch.int32).item()), 'xcc of blocks 0..15', xcc[:16], 'unit->xcc sets', [sorted(set(xcc[u::8])) for u in range(8)])
print('repoll counts fwd/bwd', ws[800:808].view(torch.int32).cpu().numpy())
st = ws[4*320:4*(320+64)].view(torch.int32).cpu().numpy().astype(np.int64)
for pas, name in ((0,'fwd'),(1,'bwd')):
    wal = st[32*pas:32*pas+(6 if pas==0 else 7)]
    print(name, 'phase times (ns)', [int((wal[i+1]-wal[i]) & 0xffffffff)*10 for i in range(len(wal)-1)])
    if pas == 1: print('   bwd inner (ns): product end -> resets drained %d, -> barrier passed %d, -> publish issued %d, -> prefetch claimed %d' % tuple(int((st[32 + b] - st[32 + a]) & 0xffffffff) * 10 for a, b in ((4, 7), (7, 8), (8, 9), (9, 5))))
    if pas == 0: print('   fwd loads issued +%d, k-step 0 valid +%d, last k-step valid +%d, product done +%d (ns after step start)' % tuple(int((st[i]-st[0]) & 0xffffffff)*10 for i in (22,23,24,2)))
    if pas == 0 and st[27] > 0: print('   fwd effective shader clock %.3f GHz' % (st[26] / (st[27] * 10.0)))
    if pas == 0: print('   fwd poll rounds %d, first round took %d ns' % (st[20], int((st[21]-st[0]) & 0xffffffff)*10))
    if pas == 0 and 0: print('   fwd inner: stamp1->6 (prefetch issue) %d, 6->7 (LDS reads + FMAs) %d, 7->2 (quad reduce) %d' % tuple(int((st[b]-st[a]) & 0xffffffff)*10 for a,b in ((1,6),(6,7),(7,2))))
