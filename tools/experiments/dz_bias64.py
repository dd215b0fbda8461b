"""dz of the forward-direction cell against a float64 autograd layer, element by element: persistent and step-wise
kernels, mean signed error per gate, by tenth of T and by |tanh gate| bucket.  usage: dz_bias64.py B T D H"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from nabu_amd import ops
from test_hip_fullsize import _layer_case
B, T, D, H = (int(a) for a in sys.argv[1:5])
lens, x, p, dout = _layer_case(B, T, D, H, seed=77, ragged=False)
ops.set_gemm_precision('f32')


def run(mode):
    plan = ops.BlstmPlan(B, T, D, H, T, mode)
    ld = torch.full((B,), T, dtype=torch.int32, device='cuda')
    out = torch.zeros((B, T, 2 * H), device='cuda')
    reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
    ops.blstm_fwd(plan, x, ld, p['fw_kernel'], p['fw_bias'], p['bw_kernel'], p['bw_bias'], out, reserve)
    n = B * T * 4 * H
    acts = reserve[:n * 4].view(torch.float32).view(B, T, 4 * H).double().clone()
    g = {k: torch.zeros_like(v) for k, v in p.items()}
    dx = torch.zeros_like(x) if D >= 256 else None
    ops.blstm_bwd(plan, x, ld, p['fw_kernel'], p['bw_kernel'], out, dout, reserve, dx, g['fw_kernel'], g['fw_bias'], g['bw_kernel'], g['bw_bias'])
    ops.check_persist_status()
    return acts, reserve[:n * 4].view(torch.float32).view(B, T, 4 * H).double().clone()


acts_p, dz_p = run(ops.LSTM_PERSISTENT)
acts_s, dz_s = run(ops.LSTM_STEPWISE)
# float64 forward + hand-written backward of the forward-direction cell (all rows full length)
xd = x.double()
K = p['fw_kernel'].double(); bvec = p['fw_bias'].double()
Wh = K[D:]
h = xd.new_zeros(B, H); c = xd.new_zeros(B, H)
A, C = [], []
for t in range(T):
    z = torch.cat([xd[:, t], h], 1) @ K + bvec
    i, j, f, o = z.split(H, 1)
    gi, gj, gf, go = torch.sigmoid(i), torch.tanh(j), torch.sigmoid(f + 1.0), torch.sigmoid(o)
    cprev = c
    c = c * gf + gi * gj
    h = torch.tanh(c) * go
    A.append((gi, gj, gf, go, cprev, c))
dz64 = xd.new_zeros(B, T, 4 * H)
dh = xd.new_zeros(B, H); dc = xd.new_zeros(B, H)
for t in range(T - 1, -1, -1):
    gi, gj, gf, go, cprev, c = A[t]
    tc = torch.tanh(c)
    dht = dout[:, t, :H].double() + dh
    dct = dc + dht * go * (1 - tc * tc)
    d = torch.cat([dct * gj * gi * (1 - gi), dct * gi * (1 - gj * gj), dct * cprev * gf * (1 - gf), dht * tc * go * (1 - go)], 1)
    dz64[:, t] = d
    dc = dct * gf
    dh = d @ Wh.t()
gj64 = torch.stack([a[1] for a in A], 1)
for name, dz, acts in (('persistent', dz_p, acts_p), ('step-wise', dz_s, acts_s)):
    print(name)
    for gate in range(4):
        e = dz[:, :, gate * H:(gate + 1) * H] - dz64[:, :, gate * H:(gate + 1) * H]
        ref = dz64[:, :, gate * H:(gate + 1) * H]
        print('  gate %d: sum of errors %+.3e (sum of dz %.3e), mean rel (e.ref/ref.ref) %+.2e, by tenth of T: %s' % (
            gate, float(e.sum() / H), float(ref.sum().abs() / H), float((e * ref).sum() / (ref * ref).sum()),
            ' '.join('%+.1e' % float(e[:, k * T // 10:(k + 1) * T // 10].sum() / H) for k in range(10))))
    ea = acts[:, :, H:2 * H] - gj64
    print('  stored tanh gate: mean error %+.2e, mean error x sign %+.2e; by |g| bucket (mean err x sign):' % (float(ea.mean()), float((ea * torch.sign(gj64)).mean())),
          ' '.join('[%.2f,%.2f) %+.1e' % (lo, hi, float((ea * torch.sign(gj64))[(gj64.abs() >= lo) & (gj64.abs() < hi)].mean())) for lo, hi in ((0, .3), (.3, .7), (.7, .95), (.95, .999), (.999, 1.01))))
    e = dz[:, :, H:2 * H] - dz64[:, :, H:2 * H]
    print('  gate-1 dz error sum by |g| bucket:', ' '.join('[%.2f,%.3f) %+.1e' % (lo, hi, float(e[(gj64.abs() >= lo) & (gj64.abs() < hi)].sum() / H)) for lo, hi in ((0, .3), (.3, .7), (.7, .95), (.95, .999), (.999, 1.01))))
print('consistency of gates 0 and 1 inside each path (same dct): sum of  dz1 gj (1 - gi) - dz0 (1 - gj^2)  from the STORED activations')
for name, dz, acts in (('persistent', dz_p, acts_p), ('step-wise', dz_s, acts_s)):
    gi, gj = acts[:, :, :H], acts[:, :, H:2 * H]
    r = dz[:, :, H:2 * H] * gj * (1 - gi) - dz[:, :, :H] * (1 - gj * gj)
    print('  %-10s sum %+.3e   sum |.| %.3e' % (name, float(r.sum() / H), float(r.abs().sum() / H)))
