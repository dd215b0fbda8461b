"""Where does the persistent backward recurrence's dz differ SYSTEMATICALLY from the step-wise kernels'?  Same layer,
both paths, dz read back from the reserve: per block of frames the relative scaling (dz_p - dz_s).dz_s / dz_s.dz_s and
the rms difference.  usage: python tools/experiments/dz_bias.py B T D H"""
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
    ld = torch.tensor(np.asarray(lens), dtype=torch.int32, device='cuda')
    out = torch.zeros((B, T, 2 * H), device='cuda')
    reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')
    ops.blstm_fwd(plan, x, ld, p['fw_kernel'], p['fw_bias'], p['bw_kernel'], p['bw_bias'], out, reserve)
    g = {k: torch.zeros_like(v) for k, v in p.items()}
    dx = torch.zeros_like(x) if D >= 256 else None
    ops.blstm_bwd(plan, x, ld, p['fw_kernel'], p['bw_kernel'], out, dout, reserve, dx, g['fw_kernel'], g['fw_bias'], g['bw_kernel'], g['bw_bias'])
    ops.check_persist_status()
    n = B * T * 4 * H
    return out, reserve[:2 * n * 4].view(torch.float32).view(2, B, T, 4 * H).double().clone(), g


out_p, dz_p, g_p = run(ops.LSTM_PERSISTENT)
out_s, dz_s, g_s = run(ops.LSTM_STEPWISE)
print('out diff rms %.3e' % float((out_p - out_s).double().pow(2).mean().sqrt()))
nb = 10
for d in range(2):
    for gate in range(4):
        a, b = dz_p[d, :, :, gate * H:(gate + 1) * H], dz_s[d, :, :, gate * H:(gate + 1) * H]
        row = []
        for i in range(nb):
            sl = slice(i * T // nb, (i + 1) * T // nb)
            da, bb = (a[:, sl] - b[:, sl]), b[:, sl]
            row.append('%+.1e/%.0e' % (float((da * bb).sum() / (bb * bb).sum()), float(da.pow(2).mean().sqrt() / bb.pow(2).mean().sqrt())))
        print('dir %d gate %d  scaling/rms per tenth of T: %s' % (d, gate, ' '.join(row)))
    print('dir %d bias grad: sum dz_p %s vs sum dz_s (first 4 cols): %s | %s' % (d, '', dz_p[d].sum((0, 1))[:4].tolist(), dz_s[d].sum((0, 1))[:4].tolist()))
