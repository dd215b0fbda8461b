import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from nabu_amd import ops
from test_hip_fullsize import _layer, _layer_case, _blstm_float64
B, T, D, H = (int(a) for a in sys.argv[1:5])
lens, x, p, dout = _layer_case(B, T, D, H, seed=77)
ops.set_gemm_precision('f32')
out_p, _, _ = _layer(B, T, D, H, lens, ops.LSTM_PERSISTENT, x, p, dout, True)
out_s, _, _ = _layer(B, T, D, H, lens, ops.LSTM_STEPWISE, x, p, dout, True)
ref, _ = _blstm_float64(x, lens, p, dout)
for name, o in (('persistent', out_p), ('step-wise', out_s)):
    e = (o.double() - ref)
    m = ref.abs() > 1e-3
    rel = (e[m] / ref[m])
    print('%-10s mean relative error %+.3e  rms %.3e  | by |h| bucket:' % (name, float(rel.mean()), float(rel.pow(2).mean().sqrt())), end=' ')
    for lo, hi in ((1e-3, 1e-2), (1e-2, 0.1), (0.1, 0.3), (0.3, 0.6), (0.6, 1.0)):
        mm = (ref.abs() >= lo) & (ref.abs() < hi)
        print('[%.0e,%.1f) %+.2e' % (lo, hi, float((e[mm] / ref[mm]).mean())), end=' ')
    print()
    # by time (forward direction half of the units)
    ef = (e[:, :, :H] * torch.sign(ref[:, :, :H]))
    print('   signed-toward-magnitude abs error by tenth of T (fw units):', ' '.join('%+.1e' % float(ef[:, i * T // 10:(i + 1) * T // 10].mean()) for i in range(10)))
