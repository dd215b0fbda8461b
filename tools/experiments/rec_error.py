"""Error of the persistent recurrence against a float64 layer, relative to the step-wise fp32 kernels' error, on one
synthetic layer: python tools/experiments/rec_error.py B T D H   (environment switches select the kernel family)."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from nabu_amd import ops
from test_hip_fullsize import _layer, _layer_case, _blstm_float64

B, T, D, H = (int(a) for a in sys.argv[1:5])
lens, x, p, dout = _layer_case(B, T, D, H, seed=77)
scale = float(os.environ.get('EXP_WSCALE', '1'))
p = {k: v * scale for k, v in p.items()}
ops.set_gemm_precision('f32')
need_dx = D >= 256
out_p, dx_p, g_p = _layer(B, T, D, H, lens, ops.LSTM_PERSISTENT, x, p, dout, need_dx)
out_s, dx_s, g_s = _layer(B, T, D, H, lens, ops.LSTM_STEPWISE, x, p, dout, need_dx)
ref, gref = _blstm_float64(x, lens, p, dout)
rms = lambda a, b: float((a.double() - b).pow(2).mean().sqrt())
print('env', {k: v for k, v in os.environ.items() if k.startswith('NABU_')}, (B, T, D, H))
print('out ratio %.3f' % (rms(out_p, ref) / rms(out_s, ref)))
for k in g_p:
    print('%-10s ratio %.3f   (x rows %.3f, h rows %.3f)' % ((k, rms(g_p[k], gref[k]) / rms(g_s[k], gref[k])) + (
        (rms(g_p[k][:D], gref[k][:D]) / rms(g_s[k][:D], gref[k][:D]), rms(g_p[k][D:], gref[k][D:]) / rms(g_s[k][D:], gref[k][D:]))
        if k.endswith('kernel') else (0, 0))))
sc = lambda a, b: float(((a.double() - b) * b).sum() / (b * b).sum())
print('systematic scaling against float64 (persistent | step-wise):')
print('  out        %+.2e | %+.2e' % (sc(out_p, ref), sc(out_s, ref)))
for k in g_p:
    print('  %-10s %+.2e | %+.2e' % (k, sc(g_p[k], gref[k]), sc(g_s[k], gref[k])))
for k in ('fw_bias', 'bw_bias'):
    ep, es = (g_p[k].double() - gref[k]), (g_s[k].double() - gref[k])
    print(k, 'per gate rms error persistent | step-wise | mean signed error persistent | step-wise | rms of the gradient')
    for gate in range(4):
        sl = slice(gate * H, (gate + 1) * H)
        print('   gate %d  %.3e | %.3e | %+.3e | %+.3e | %.3e' % (gate, float(ep[sl].pow(2).mean().sqrt()), float(es[sl].pow(2).mean().sqrt()),
              float(ep[sl].mean()), float(es[sl].mean()), float(gref[k][sl].pow(2).mean().sqrt())))
