"""Helpers of tests/test_hip_real_operands.py and tools/experiments/real_operands.py: the dense products of a BLSTM
layer as (name, A [M, K], B [N, K]) pairs with C = A . B^T — exactly the operand matrices nabu_blstm_fwd / _bwd multiply
(nabu_amd/csrc/lstm.hip) — and their error against float64 under each arithmetic."""
import numpy as np
import torch


def layer_products(c):
    """c: one entry of layer.CAPTURE (x [B,T,D], out [B,T,2H], kf/kb [(D+H),4H], dz [2, BT, 4H])"""
    B, T, D, H = c['B'], c['T'], c['D'], c['H']
    BT, G = B * T, 4 * H
    x = c['x'].reshape(BT, D)
    dz = c['dz']                                             # [2, BT, G]
    wx = torch.cat([c['kf'][:D], c['kb'][:D]], dim=1)        # [D, 2G]
    dzc = torch.cat([dz[0], dz[1]], dim=1)                   # [BT, 2G]
    prods = []
    if D >= 256:
        prods.append(('fwd x.Wx', x, wx.t().contiguous()))                         # [BT, D] x [2G, D]
        prods.append(('dx dz.Wx^T', dzc, wx.contiguous()))                         # [BT, 2G] x [D, 2G]
        prods.append(('dWx x^T.dz', x.t().contiguous(), dzc.t().contiguous()))     # [D, BT] x [2G, BT]
    out = c['out']
    for d, name in ((0, 'fw'), (1, 'bw')):
        h = out[:, :, d * H:(d + 1) * H]
        hp = torch.zeros_like(h)
        if d == 0:
            hp[:, 1:] = h[:, :-1]
        else:
            hp[:, :-1] = h[:, 1:]
        a = hp.reshape(BT, H)
        if D < 256:                                          # narrow input: the whole kernel gradient is one product
            a = torch.cat([x, a], dim=1)
        prods.append(('dW%s %s^T.dz' % ('' if D < 256 else 'h', name), a.t().contiguous(), dz[d].t().contiguous()))
    return prods


def _pk(a, b, planes, direct, bound_a=None, bound_b=None):
    from nabu_amd import ops
    M, K = a.shape
    N = b.shape[0]
    pa, pb = ops.PackedOperand(M, K, planes, 'cuda'), ops.PackedOperand(N, K, planes, 'cuda')
    if planes == 2:
        ops.pk_pack(pa, a, bound=bound_a)
        ops.pk_pack(pb, b, bound=bound_b)
    else:
        ops.pk_pack(pa, a)
        ops.pk_pack(pb, b)
    c = torch.empty(M, N, device='cuda')
    ops.gemm_pk(pa, pb, c, planes, direct=direct)
    return c


def product_errors(a, b):
    """rms and max error against float64 of every arithmetic, relative to the exact-fp32 MFMA kernel's"""
    from nabu_amd import ops
    ref = a.double() @ b.double().t()
    c32 = torch.empty(a.shape[0], b.shape[0], device='cuda')
    ops.gemm(a, b, c32, False, True, precision='f32')
    e32 = (c32.double() - ref)
    r32, m32 = float(e32.pow(2).mean().sqrt()), float(e32.abs().max())
    del c32, e32
    res = {'fp32': (r32, m32)}
    for name, planes, direct in (('f16x3', 2, 2), ('bf16x6', 3, 0)):
        e = _pk(a, b, planes, direct).double() - ref
        res[name] = (float(e.pow(2).mean().sqrt()) / r32, float(e.abs().max()) / m32)
        del e
    return res
