"""Helper of tests/test_hip_fullsize.py::test_alternative_persistent_kernel_families: one BLSTM layer forward + backward
on the persistent kernels the environment of THIS process selects (NABU_PERSIST_MX, NABU_PERSIST_MXF, NABU_PERSIST_FUSE_INPUT are read once per
process), against the step-wise fp32 kernels; prints 'FAMILY OK <largest relative error>'.
usage: python tests/persist_family_check.py B T D H"""
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
sys.path.insert(0, here)
import torch  # noqa: E402

from nabu_amd import ops  # noqa: E402
from test_hip_fullsize import _layer, _layer_case, _rel  # noqa: E402

B, T, D, H = (int(a) for a in sys.argv[1:5])
lens, x, p, dout = _layer_case(B, T, D, H, seed=B + T + H)
need_dx = D != 40
out_p, dx_p, g_p = _layer(B, T, D, H, lens, ops.LSTM_PERSISTENT, x, p, dout, need_dx)
out_s, dx_s, g_s = _layer(B, T, D, H, lens, ops.LSTM_STEPWISE, x, p, dout, need_dx)
ops.check_persist_status()
assert bool(torch.isfinite(out_p).all())
worst = float((out_p - out_s).abs().max())
assert worst < 2e-6, worst
for k in g_p:
    r = _rel(g_p[k], g_s[k])
    assert r < 3e-6, (k, r)
    worst = max(worst, r)
if need_dx:
    r = _rel(dx_p, dx_s)
    assert r < 3e-6, ('dx', r)
    worst = max(worst, r)
print('FAMILY OK %.3e' % worst)
