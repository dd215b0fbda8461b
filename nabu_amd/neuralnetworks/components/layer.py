"""Neural network layers (reference: nabu/neuralnetworks/components/layer.py).

Same functions, same variable names; the body of ``blstm`` is ONE call into the
C ABI (nabu_blstm_fwd) instead of two tf.while_loops of ~15 small ops per step."""
import torch

from nabu_amd import ops as hip
from nabu_amd import variables as vs
from nabu_amd.autodiff import record, requires_grad, SeqLen, Tape
from nabu_amd.neuralnetworks.components import ops

# cell scope of tf.contrib.rnn.LayerNormBasicLSTMCell under bidirectional_dynamic_rnn
_CELL = 'bidirectional_rnn/%s/layer_norm_basic_lstm_cell'
LSTM_MODE = [hip.LSTM_AUTO]      # tests flip this to compare the two recurrent paths
# arithmetic of the input-to-hidden GEMMs of the layers built next ('default' | 'f32' | 'bf16' |
# 'bf16x3' | 'bf16x6'); set by the encoders from their `gemm_precision` cfg key
GEMM_PRECISION = ['default']
# arithmetic of the recurrent product h.W_h of the layers built next ('default' = three fp16 plane products of row-scaled
# operands, fp32-equivalent | 'f32' = the exact-fp32 kernels); set by the encoders from their `recurrent_precision` key
RECURRENT_PRECISION = ['default']
# the weight-gradient products of a layer run after the LAST recurrence of the backward pass (Tape.defer): nothing
# waits for them, and bf16 matrix bursts in front of a persistent recurrent kernel slow it down (include/nabu_hip.h,
# nabu_blstm_bwd_data).  NABU_DEFER_WGRAD=0: one fused nabu_blstm_bwd per layer as before.
import os as _os
DEFER_WEIGHT_GRADS = [_os.environ.get('NABU_DEFER_WGRAD', '1') != '0']
# tests: a list here receives, per layer and backward pass, the operands of the layer's dense products (x, out, the
# kernels and a copy of dz as the data part of the backward pass left it in the reserve) — tests/test_hip_real_operands.py
CAPTURE = [None]


# packed companions (include/nabu_hip.h, ABI version 3): the forward recurrent kernel writes the layer's output also as the
# next layer's f16x3 operands and its own h^T operand.  NABU_PACKED_COMPANIONS=0: every call packs for itself as before.
PACKED_COMPANIONS = [_os.environ.get('NABU_PACKED_COMPANIONS', '1') != '0']
# NABU_CHECK_X_BOUND=1: every layer call measures max|x| (host round trip) and raises if it exceeds the bound it was promised
CHECK_X_BOUND = [_os.environ.get('NABU_CHECK_X_BOUND', '0') == '1']


def blstm(inputs, sequence_length, num_units, layer_norm=False, scope=None, out_stack=0):
    """A BLSTM layer (reference layer.py:8-51).

    inputs [B,T,D] fp32 contiguous on the GPU; sequence_length [B];
    returns [B,T,2*num_units] = concat(fw, bw).  Variables (TF layout):
    <scope>/bidirectional_rnn/{fw,bw}/layer_norm_basic_lstm_cell/{kernel [(D+H),4H], bias [4H]},
    gate order i,j,f,o; both use the scope-default glorot-uniform initialiser."""
    if layer_norm:
        raise NotImplementedError('layer_norm=True is not on the hot path (the reference '
                                  'always calls blstm with layer_norm=False)')
    lens = SeqLen.wrap(sequence_length, inputs.device)
    B, T, D = inputs.shape
    H = int(num_units)
    with vs.variable_scope(scope or 'BLSTM'):
        kf = vs.get_variable((_CELL % 'fw') + '/kernel', [D + H, 4 * H])
        bf = vs.get_variable((_CELL % 'fw') + '/bias', [4 * H])
        kb = vs.get_variable((_CELL % 'bw') + '/kernel', [D + H, 4 * H])
        bb = vs.get_variable((_CELL % 'bw') + '/bias', [4 * H])
    # what is known about the input's magnitude (ops.value_bound: the previous layer's outputs, |o tanh c| <= 1, through
    # pyramid stacking and dropout) spares the f16x3 packs of x their measuring pass; no tape = no backward pass: the
    # reserve then holds the activations only
    training = Tape.current is not None
    plan = hip.BlstmPlan(B, T, D, H, min(lens.max(), T), LSTM_MODE[0], GEMM_PRECISION[0], x_bound=ops.value_bound(inputs),
                         fwd_only=not training, recurrent_precision=RECURRENT_PRECISION[0], out_stack=out_stack)
    x = inputs if inputs.is_contiguous() else inputs.contiguous()
    if CHECK_X_BOUND[0] and ops.value_bound(inputs) > 0:
        # debug mode (NABU_CHECK_X_BOUND=1): the recorded bound is a PROMISE (ops.set_value_bound) — an in-place change
        # of a bounded tensor would leave it stale and the f16x3 packs would overflow to Inf without a diagnostic
        worst = float(x.detach().cpu().numpy().__abs__().max())
        if not worst <= ops.value_bound(inputs) * (1 + 1e-6):
            raise RuntimeError('blstm input exceeds its recorded bound: max|x| = %g > %g (a tensor carrying a value bound '
                               'was modified in place?)' % (worst, ops.value_bound(inputs)))
    out_pk = None
    if PACKED_COMPANIONS[0]:
        where = (vs.current_scope(), scope or 'BLSTM', B, T, D, H)
        # input: the producer layer's kernel wrote it (ops.packed), if this layer's products read packed operands at all
        x_pk = ops.packed(inputs, 1) if x is inputs else None
        if x_pk is not None and plan.pk_bytes[0] and (x_pk[0].numel(), x_pk[1].numel()) == tuple(plan.pk_bytes[:2]):
            hip.blstm_set_companions(plan, x_pk=x_pk)
        # output, for the consumer behind `out_stack` stacked frames: only if that layer (same units, same arithmetic)
        # would read them
        if out_stack in (1, 2) and T % out_stack == 0 and plan.pk_bytes[3]:
            nxt = hip.BlstmPlan(B, T // out_stack, 2 * H * out_stack, H, T // out_stack, LSTM_MODE[0], GEMM_PRECISION[0], x_bound=1.0,
                                fwd_only=not training, recurrent_precision=RECURRENT_PRECISION[0])
            if nxt.pk_bytes[0] == plan.pk_bytes[3] and nxt.pk_bytes[1] == plan.pk_bytes[4]:
                out_pk = (ops.resident_zeros(where + ('rows',), plan.pk_bytes[3], x.device, Tape.current),
                          ops.resident_zeros(where + ('cols',), plan.pk_bytes[4], x.device, Tape.current))
                hip.blstm_set_companions(plan, out_pk=out_pk)
        if training and plan.pk_bytes[2]:
            hip.blstm_set_companions(plan, hT_pk=ops.resident_zeros(where + ('hT',), plan.pk_bytes[2], x.device, Tape.current))
        # what the recurrent kernel does not write itself is packed where it is needed, as before (no companion)
        mask = hip.blstm_emits_packed(plan)
        if out_pk is not None and (mask & 3) != 3:
            hip.blstm_drop_companions(plan, out_pk=True)
            out_pk = None
        if not mask & 4:
            hip.blstm_drop_companions(plan, hT_pk=True)
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=x.device)
    reserve = torch.empty(plan.reserve_bytes, dtype=torch.uint8, device=x.device)
    hip.blstm_fwd(plan, x, lens.dev, kf.data, bf.data, kb.data, bb.data, out, reserve)
    ops.set_value_bound(out, 1.0)
    if out_pk is not None:
        ops.set_packed(out, out_stack, out_pk)
    need_dx = requires_grad(inputs)

    def backward(dout):
        for v in (kf, bf, kb, bb):
            if v.grad is None:
                v.grad = torch.zeros_like(v.data)
        dx = torch.empty_like(x) if need_dx else None
        tape = Tape.current_backward
        if DEFER_WEIGHT_GRADS[0] and tape is not None:
            hip.blstm_bwd_data(plan, x, lens.dev, kf.data, kb.data, out, dout.contiguous(), reserve, dx, bf.grad, bb.grad)
            if CAPTURE[0] is not None:
                n = B * T * 4 * H
                dz = reserve[:2 * n * 4].view(torch.float32).view(2, B * T, 4 * H).clone()
                CAPTURE[0].append({'x': x, 'out': out, 'kf': kf.data, 'kb': kb.data, 'dz': dz, 'B': B, 'T': T, 'D': D, 'H': H,
                                   'bf': bf.data, 'bb': bb.data, 'lens': lens.dev.cpu().numpy(), 'dout': dout.contiguous().clone()})
            tape.defer(lambda: hip.blstm_bwd_weights(plan, x, lens.dev, out, reserve, kf.grad, kb.grad), params=(kf, kb))
        else:
            hip.blstm_bwd(plan, x, lens.dev, kf.data, kb.data, out, dout.contiguous(), reserve, dx,
                          kf.grad, bf.grad, kb.grad, bb.grad)
        return [dx]
    record([inputs], [out], backward, params=(kf, bf, kb, bb))
    return out


def pblstm(inputs, sequence_length, num_units, num_steps=2, layer_norm=False, scope=None):
    """A pyramidal BLSTM layer (reference layer.py:53-94): blstm, then stack
    ``num_steps`` consecutive output frames.  Returns (outputs, new lengths)."""
    with vs.variable_scope(scope or 'PBLSTM'):
        outputs = blstm(inputs=inputs, sequence_length=sequence_length, num_units=num_units,
                        layer_norm=layer_norm, out_stack=num_steps if num_steps in (1, 2) else 0)
        outputs, output_seq_lengths = ops.pyramid_stack(outputs, sequence_length, num_steps)
    return outputs, output_seq_lengths
