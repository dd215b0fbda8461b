"""Persistent recurrence, bf16-plane product (lstm_persist_mx.hip) against the step-wise kernels and against the
exact-fp32 persistent kernels (NABU_PERSIST_MX=0, separate process), with timing per sequential step.
usage: python tools/experiments/mx_check.py [B T D H]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from nabu_amd import ops
sys.path.insert(0, 'tests')
from test_hip_fullsize import _layer, _layer_case, _rel

B, T, D, H = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 500, 2048, 512)
lens, x, p, dout = _layer_case(B, T, D, H, seed=T + D)
need_dx = D != 40
out_p, dx_p, g_p = _layer(B, T, D, H, lens, ops.LSTM_PERSISTENT, x, p, dout, need_dx)
out_s, dx_s, g_s = _layer(B, T, D, H, lens, ops.LSTM_STEPWISE, x, p, dout, need_dx)
print('MX', os.environ.get('NABU_PERSIST_MX', '1'), 'shape', (B, T, D, H))
print('out max abs diff', float((out_p - out_s).abs().max()), 'finite', bool(torch.isfinite(out_p).all()))
for k in g_p:
    print(k, 'rel', _rel(g_p[k], g_s[k]))
if need_dx:
    print('dx rel', _rel(dx_p, dx_s))
# timing
plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_PERSISTENT)
ld = torch.full((B,), T, dtype=torch.int32, device='cuda')
out = torch.zeros(B, T, 2 * H, device='cuda'); reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
g = {k: torch.zeros_like(v) for k, v in p.items()}; dx = torch.zeros_like(x)
prof = ops.enable_profiler()
for it in range(6):
    ops.blstm_fwd(plan, x, ld, p['fw_kernel'], p['fw_bias'], p['bw_kernel'], p['bw_bias'], out, reserve)
    ops.blstm_bwd(plan, x, ld, p['fw_kernel'], p['bw_kernel'], out, dout, reserve, dx, g['fw_kernel'], g['fw_bias'], g['bw_kernel'], g['bw_bias'])
torch.cuda.synchronize()
ops.check_persist_status()
recs = prof.collect()
print(' '.join('%s %.3f us/step' % (r[0], r[4] * 1e3 / T) for r in recs[4:]))
