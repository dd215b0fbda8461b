# half-batch pipeline feasibility: B=16 persistent recurrence (one workgroup per CU) beside GEMMs on a side stream
import sys, os, time; sys.path.insert(0,'.')
import numpy as np, torch
from nabu_amd import ops, _hip
B,T,D,H = 16,500,2048,512
x = torch.randn(B,T,D, device='cuda')*0.1
lens = torch.full((B,), T, dtype=torch.int32).cuda()
p = [torch.randn(s, device='cuda')*0.03 for s in [(D+H,4*H),(4*H,),(D+H,4*H),(4*H,)]]
dout = torch.randn(B,T,2*H, device='cuda')
plan = ops.BlstmPlan(B,T,D,H,T,ops.LSTM_PERSISTENT)
out = torch.zeros(B,T,2*H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = [torch.zeros_like(q) for q in p]; dx = torch.zeros_like(x)
# GEMMs of the OTHER half at this layer size: dx (NT 8000x2048x2048), dWx (TN 2048x2048x8000)
M = 8000
a_nt = torch.randn(M, 2048, device='cuda'); w_nt = torch.randn(2048, 2048, device='cuda'); c_nt = torch.zeros(M, 2048, device='cuda')
a_tn = torch.randn(M, 2048, device='cuda'); b_tn = torch.randn(M, 2048, device='cuda'); c_tn = torch.zeros(2048, 2048, device='cuda')
ws2 = torch.zeros(128 << 20, dtype=torch.uint8, device='cuda')
L = _hip.lib()
def gemm_side(stream, n=2):
    for _ in range(n):
        _hip.check(L.nabu_gemm_ex(1, 0, 1, M, 2048, 2048, 1.0, a_nt.data_ptr(), 2048, w_nt.data_ptr(), 2048, 0.0, c_nt.data_ptr(), 2048, None, 0, 0, 0, ws2.data_ptr(), ws2.numel(), stream.cuda_stream), 'gemm')
        _hip.check(L.nabu_gemm_ex(1, 1, 0, 2048, 2048, M, 1.0, a_tn.data_ptr(), 2048, b_tn.data_ptr(), 2048, 0.0, c_tn.data_ptr(), 2048, None, 0, 0, 0, ws2.data_ptr(), ws2.numel(), stream.cuda_stream), 'gemm')
prof = ops.enable_profiler()
s_hi = torch.cuda.Stream(priority=-1); s2 = torch.cuda.Stream(priority=0)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
def rec_fwd(): ops.blstm_fwd(plan, x, lens, p[0],p[1],p[2],p[3], out, reserve)
rec_fwd(); torch.cuda.synchronize()
# the fwd call = input GEMMs + recurrence; time the recurrence alone through the profiler records
t_fwd = timed(rec_fwd)
t_gemm = timed(lambda: gemm_side(torch.cuda.current_stream()))
def both():
    with torch.cuda.stream(s_hi): rec_fwd()
    gemm_side(s2)
t_both = timed(both)
recs = prof.collect()
print('B=16 T=500: blstm_fwd (2 input GEMMs + recurrence) alone %.2f ms, side GEMMs (2x NT + 2x TN at M=8000) alone %.2f ms, sum %.2f, concurrent %.2f ms' % (t_fwd, t_gemm, t_fwd + t_gemm, t_both))
print('recurrent kernel ms per launch (alone ... concurrent):', ['%.2f' % r[4] for r in recs])
ops.check_persist_status()
