"""Spread of publish / poll-done times inside one exchange step of the forward recurrence (NABU_PERSIST_DEBUG=4096)."""
import sys, os; sys.path.insert(0,'.')
os.environ.setdefault('NABU_PERSIST_DEBUG', '4096')
import numpy as np, torch
from nabu_amd import ops, _hip
B,T,D,H = 32,500,2048,512
x = torch.randn(B,T,D, device='cuda')*0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda')*0.03 for s in [(D+H,4*H),(4*H,),(D+H,4*H),(4*H,)]]
plan = ops.BlstmPlan(B,T,D,H,T,ops.LSTM_PERSISTENT)
out = torch.zeros(B,T,2*H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
BWD = len(sys.argv) > 1 and sys.argv[1] == 'bwd'
dout = torch.randn(B,T,2*H, device='cuda'); g = [torch.zeros_like(q) for q in p]; dx = torch.zeros_like(x)
for it in range(2):
    ops.blstm_fwd(plan, x, lens, p[0],p[1],p[2],p[3], out, reserve)
    if BWD: ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0],g[1],g[2],g[3])
torch.cuda.synchronize()
print('backward' if BWD else 'forward')
ws = _hip.Workspace._bufs[(str(x.device),'blstm')]
st = ws[4*384:4*(384+128)].view(torch.int32).cpu().numpy().astype(np.int64).reshape(2,32,2)
for u,name in ((0,'unit 0'),(1,'second unit')):
    pub = st[u,:,0] & 0x0fffffff; t0 = pub.min()
    d = sorted(((pub - t0))*10)
    pdl = ((st[u,:,1] & 0x0fffffff) - t0)*10
    pd = sorted(pdl)
    print(name, 'publish spread: median %d p90 %d max %d ns | last wave poll-done after first publish: min %d median %d max %d' % (d[16], d[28], d[31], pd[0], pd[16], pd[31]))
    print('   publish (ns)', ((pub - t0)*10).tolist())
    print('   poll done  ', pdl.tolist())
print('offset between the two units first publish (ns):', int((((st[1,:,0]&0x0fffffff).min()-(st[0,:,0]&0x0fffffff).min()))*10))
