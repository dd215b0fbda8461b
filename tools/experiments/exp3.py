import sys, os, time; sys.path.insert(0,'.')
import numpy as np, torch
from nabu_amd import ops, _hip
B,T,D,H = 32,500,2048,512
x = torch.randn(B,T,D, device='cuda')*0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda')*0.03 for s in [(D+H,4*H),(4*H,),(D+H,4*H),(4*H,)]]
dout = torch.randn(B,T,2*H, device='cuda')
plan = ops.BlstmPlan(B,T,D,H,T,ops.LSTM_PERSISTENT)
out = torch.zeros(B,T,2*H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = [torch.zeros_like(q) for q in p]; dx = torch.zeros_like(x)
# a dW-like GEMM of the next-higher layer: x2^T [2048 x 16000] . dz2 [16000 x 2048]
x2 = torch.randn(16000, 2048, device='cuda'); dz2 = torch.randn(16000, 2048, device='cuda'); dw = torch.zeros(2048, 2048, device='cuda')
ws2 = torch.zeros(64 << 20, dtype=torch.uint8, device='cuda')
L = _hip.lib()
def gemm_side(stream):
    for _ in range(3):
        _hip.check(L.nabu_gemm_ex(1, 1, 0, 2048, 2048, 16000, 1.0, x2.data_ptr(), 2048, dz2.data_ptr(), 2048, 0.0, dw.data_ptr(), 2048, None, 0, 0, 0, ws2.data_ptr(), ws2.numel(), stream.cuda_stream), 'gemm')
prof = ops.enable_profiler()
s2 = torch.cuda.Stream()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
ops.blstm_fwd(plan, x, lens, p[0],p[1],p[2],p[3], out, reserve)
def rec():
    ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0],g[1],g[2],g[3])
rec(); gemm_side(torch.cuda.current_stream()); torch.cuda.synchronize()
t_rec = timed(rec); t_gemm = timed(lambda: gemm_side(torch.cuda.current_stream()))
def both():
    rec()                       # main stream: persistent kernel first (grabs its slots), then its GEMMs
    gemm_side(s2)               # side stream
t_both = timed(both)
recs = prof.collect()
print('blstm_bwd alone %.2f ms, 3 side GEMMs alone %.2f ms, serial sum %.2f ms, concurrent %.2f ms' % (t_rec, t_gemm, t_rec + t_gemm, t_both))
print('recurrent kernel ms:', ['%.2f' % r[4] for r in recs if r[0] != 'fwd'][-3:])
ops.check_persist_status()
