"""Is the forward recurrent kernel's time a function of its operands?  (zeros, small / large weights, saturated gates;
with the in-kernel clock stamps)  It is not: 1.426-1.436 us per step.  python tools/experiments/rec_data_independence.py"""
import sys, os; sys.path.insert(0,'.')
import torch
from nabu_amd import ops
def run(x, p, tag, H=512):
    B, T, D = x.shape
    lens = torch.full((B,), T, dtype=torch.int32).cuda()
    plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_PERSISTENT)
    out = torch.zeros(B, T, 2 * H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
    for it in range(3):
        ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for it in range(6):
        ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
    b.record(); torch.cuda.synchronize()
    ops.check_persist_status()
    print('%-34s call %.3f ms = %.3f us/step  clock %s  out rms %.3f |h|max %.3f' % (tag, a.elapsed_time(b) / 6, a.elapsed_time(b) / 6 * 1e3 / T, ops.persist_clocks()['fwd'], float(out.pow(2).mean().sqrt()), float(out.abs().max())), flush=True)
torch.manual_seed(0)
xr = torch.randn(32, 1000, 40, device='cuda')
pr = [torch.randn(s, device='cuda') * 0.03 for s in [(552, 2048), (2048,), (552, 2048), (2048,)]]
for rep in range(2):
    run(xr * 0.1, pr, 'XIN x*0.1, W*0.03')
    run(xr, pr, 'XIN x*1, W*0.03')
    run(xr, [w * 3 for w in pr], 'XIN x*1, W*0.09')
    run(xr * 3, [w * 10 for w in pr], 'XIN x*3, W*0.3 (saturated)')
    run(xr * 0, [w * 0 for w in pr], 'XIN zeros')
x1 = torch.randn(32, 500, 2048, device='cuda')
p1 = [torch.randn(s, device='cuda') * 0.03 for s in [(2560, 2048), (2048,), (2560, 2048), (2048,)]]
run(x1 * 0.1, p1, 'L1 x*0.1, W*0.03')
run(x1, [w * 3 for w in p1], 'L1 x*1, W*0.09')
run(x1 * 0, [w * 0 for w in p1], 'L1 zeros')
