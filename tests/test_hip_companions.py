"""Packed companions of a BLSTM layer's output (include/nabu_hip.h, nabu_blstm_desc ABI version 3): the forward
recurrent kernel writes `out` ALSO as the next layer's f16x3 operands (rows and transposed) and as its own h_(t-1)^T
operand.  What the kernel writes must be, bit for bit, what the pack kernels make of the `out` it stored
(reference semantics: layer.blstm + ops.pyramid_stack, components/layer.py:8-94, components/ops.py:6-60 — the packs are
operand FORMATS of the dense products, not values of the reference graph), and a model that uses them must train like
one that packs for itself."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ONE = 0x3F800000


def _case(B, T, D, H, lens, seed):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    lens = np.asarray(lens)
    x = torch.randn((B, T, D), generator=gen, device=DEV)
    mask = (torch.arange(T, device=DEV)[None, :] < torch.tensor(lens, device=DEV)[:, None])
    x = x * mask[:, :, None]
    s = 1.0 / np.sqrt(D + H)
    p = {k: torch.randn(shape, generator=gen, device=DEV) * s
         for k, shape in (('fw_kernel', (D + H, 4 * H)), ('fw_bias', (4 * H,)), ('bw_kernel', (D + H, 4 * H)), ('bw_bias', (4 * H,)))}
    return lens, x, p


def _expected(out, B, T, H, S, hT_row0):
    """the packs the pack kernels make of `out` at the companions' scale (row maxima = 1.0f)"""
    from nabu_amd import ops
    R, C = B * T // S, 2 * H * S
    src = out.view(R, C)
    rows = ops.PackedOperand(R, C, 2, DEV)
    rows.buf.zero_()
    rows.amax.fill_(ONE)
    ops.pk_pack(rows, src, measure=False)
    cols = ops.PackedOperand(C, R, 2, DEV)
    cols.buf.zero_()
    cols.amax.fill_(ONE)
    ops.pk_pack(cols, src, transposed=True, measure=False)
    hT = []
    for d in range(2):
        o = ops.PackedOperand(hT_row0 + H, B * T, 2, DEV)
        o.buf.zero_()
        o.amax.fill_(ONE)
        ops.pk_pack(o, out.view(B * T, 2 * H)[:, d * H:(d + 1) * H], transposed=True, row_off=hT_row0, period=T,
                    shift=1 if d else -1, measure=False, R=B * T, C=H, ld=2 * H)
        hT.append(o)
    return rows, cols, hT


@pytest.mark.parametrize('B,T,D,H,S,lens,emits', [
    (32, 64, 256, 128, 2, 'ragged_full', True),       # pyramid layer, a row reaches T: the kernel writes the companions
    (32, 64, 256, 128, 1, 'ragged_full', True),       # plain stack of layers (DBLSTM): one frame per packed row
    (32, 70, 256, 256, 2, 'full', True),              # B T / 2 = 1120 rows: not a multiple of 16 — k-blocks shared between batch rows
    (24, 96, 40, 128, 2, 'ragged_full', 'switch'),    # first layer: narrow input projected inside the kernel, x^T in front of h^T:
                                                      # by default its h^T is left to the pack kernels, by the kernel when the switch says so
    (32, 64, 256, 128, 2, 'short', False),            # max(len) < T: frames the recurrence never visits -> the pack kernels
    (40, 64, 256, 256, 2, 'ragged_full', False),      # 33 .. 64 rows run as two launches: the pack kernels
])
def test_forward_kernel_writes_the_packed_companions_bit_exactly(B, T, D, H, S, lens, emits):
    from nabu_amd import ops
    rng = np.random.default_rng(B + T)
    if lens == 'full':
        lens = np.full(B, T)
    elif lens == 'short':
        lens = rng.integers(1, T - 3, B)
    else:
        lens = rng.integers(1, T + 1, B)
        lens[1] = T
        lens[2] = 0 if B > 8 else lens[2]
    lens, x, p = _case(B, T, D, H, lens, seed=B * T + S)
    plan = ops.BlstmPlan(B, T, D, H, int(max(lens)), ops.LSTM_AUTO, 'f16x3', out_stack=S)
    assert plan.pk_bytes[2] and plan.pk_bytes[3] and plan.pk_bytes[4], plan.pk_bytes
    # poisoned where the kernel must write, zero where nothing may ever be written: start from zeros (the contract) and
    # check afterwards that everything the reference packs hold is there
    out_pk = (torch.zeros(plan.pk_bytes[3], dtype=torch.uint8, device=DEV), torch.zeros(plan.pk_bytes[4], dtype=torch.uint8, device=DEV))
    hT_pk = torch.zeros(plan.pk_bytes[2], dtype=torch.uint8, device=DEV)
    ops.blstm_set_companions(plan, out_pk=out_pk, hT_pk=hT_pk)
    if emits == 'switch':
        import os
        emits = 'NABU_PERSIST_EMIT_MASK' in os.environ
    assert bool(ops.blstm_emits_packed(plan)) == emits
    ld = torch.tensor(lens, dtype=torch.int32, device=DEV)
    out = torch.full((B, T, 2 * H), float('nan'), device=DEV)
    reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device=DEV)
    for _ in range(2):      # a second call over the same buffers (as every training step does)
        ops.blstm_fwd(plan, x, ld, p['fw_kernel'], p['fw_bias'], p['bw_kernel'], p['bw_bias'], out, reserve)
    ops.check_persist_status()
    r0 = D if D < 256 else 0
    rows, cols, hT = _expected(out, B, T, H, S, r0)
    assert torch.equal(out_pk[0], rows.buf), 'rows companion differs from pack(out)'
    assert torch.equal(out_pk[1], cols.buf), 'transposed companion differs from pack(out)'
    half = plan.pk_bytes[2] // 2
    for d in range(2):
        got = hT_pk[d * half:(d + 1) * half].view(-1, 2, hT[d].rows_pad, 32)
        want = hT[d].buf.view(-1, 2, hT[d].rows_pad, 32)
        assert torch.equal(got[:, :, r0:r0 + H], want[:, :, r0:r0 + H]), 'h^T companion of cell %d differs from pack(out)' % d
        assert not got[:, :, r0 + H:].any() and not got[:, :, :r0].any()     # rows the forward pass does not own stay untouched


def test_listener_step_is_the_same_with_and_without_companions():
    """loss and every gradient of a shrunken cfg2 Listener step (every layer's products on f16x3 operands; ragged batch,
    the longest utterance fills T) with the companions written by the recurrent kernels against the same model packing
    for itself — the operands differ only in their power-of-two scale (2^14 from the kernel, 2^15 from the pack's
    bound) — and a 3-step clip + Adam loss trajectory"""
    from nabu_amd import recipes
    from nabu_amd.autodiff import Tape
    from nabu_amd.neuralnetworks.components import layer
    from nabu_amd.neuralnetworks.trainers import trainer_factory, loss_functions
    from nabu_amd.processing.synthetic import SyntheticData
    B, T, H = 32, 512, 128

    def run(on):
        layer.PACKED_COMPANIONS[0] = on
        data = SyntheticData(B, T, 40, min_frames=T // 2, min_labels=2, max_labels=6, time_reduction=8, seed=99)
        mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **{'encoder.num_units': H, 'trainer.batch_size': B})
        tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec, expdir=None,
                                                 server=None, task_index=0)
        raw = data.batch(0)
        raw['input_seq_length']['features'][0] = T        # the longest utterance fills the batch (as a padded batch does)
        batch = tr.to_device(raw)
        with Tape() as tape:
            logits, lsl = tr.model(batch['inputs'], batch['input_seq_length'], batch['targets'], batch['target_seq_length'], True)
            loss = loss_functions.CTC(batch['targets'], logits, lsl, batch['target_seq_length'])
        tape.backward(loss)
        grads = {v.name: v.grad.cpu().numpy().copy() for v in tr.model.variables}
        losses = [float(tr.step(tr.to_device(data.batch(i))).item()) for i in range(3)]
        return float(loss.item()), grads, losses
    try:
        l_on, g_on, t_on = run(True)
        l_off, g_off, t_off = run(False)
    finally:
        layer.PACKED_COMPANIONS[0] = True
    assert abs(l_on - l_off) <= 2e-6 * abs(l_off), (l_on, l_off)
    for k in g_on:
        assert np.abs(g_on[k] - g_off[k]).max() <= 1e-5 * np.abs(g_off[k]).max() + 1e-9, k
    assert np.allclose(t_on, t_off, rtol=1e-5, atol=0), (t_on, t_off)


def test_all_three_companions_out_of_the_kernel_in_their_own_process():
    """the default lets the recurrent kernel write h^T only (NABU_PERSIST_EMIT_MASK = 4, include/nabu_hip.h); the rows and
    the transposed operand out of the kernel (mask 7) stay parity-green: the bit-exactness cases above once more, in a
    process of their own (the switch is read once per process)"""
    import os
    import subprocess
    import sys
    if 'NABU_PERSIST_EMIT_MASK' in os.environ:
        pytest.skip('already a process with the switch set')
    e = dict(os.environ)
    e['NABU_PERSIST_EMIT_MASK'] = '7'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'bit_exactly or step_is_the_same'],
                       env=e, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    # ... and the switch at 4, taken literally: h^T also out of the first layer's launch (the default leaves that one to
    # the pack kernels: its stores cost the launch more than the pack they replace)
    e['NABU_PERSIST_EMIT_MASK'] = '4'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'bit_exactly and 24-96-40'],
                       env=e, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and '1 passed' in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize('size', [32, 44])
def test_older_descriptor_layouts_are_still_accepted(size):
    """ABI version 3 grew nabu_blstm_desc; a caller compiled against version 1 (32 bytes: everything up to gemm_precision —
    INTEGRATION.md's example binding) or version 2 (44 bytes: + x_bound, flags, recurrent_precision) passes its own size
    and gets the same layer: the fields behind it read as 0"""
    from nabu_amd import ops, _hip
    B, T, D, H = 8, 24, 40, 128
    lens, x, p = _case(B, T, D, H, np.full(B, T), seed=5)
    ld = torch.tensor(lens, dtype=torch.int32, device=DEV)
    outs = []
    for sz in (ctypes.sizeof(_hip.BlstmDesc), size):
        plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_AUTO)
        plan.desc.size = sz
        L = _hip.lib()
        plan.reserve_bytes = L.nabu_blstm_reserve_bytes(ctypes.byref(plan.desc))
        plan.ws_bytes = L.nabu_blstm_ws_bytes(ctypes.byref(plan.desc))
        assert plan.reserve_bytes > 0
        out = torch.full((B, T, 2 * H), float('nan'), device=DEV)
        reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device=DEV)
        ops.blstm_fwd(plan, x, ld, p['fw_kernel'], p['fw_bias'], p['bw_kernel'], p['bw_bias'], out, reserve)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    bad = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_AUTO)
    bad.desc.size = 40
    assert _hip.lib().nabu_blstm_reserve_bytes(ctypes.byref(bad.desc)) == 0      # not a layout of any version
