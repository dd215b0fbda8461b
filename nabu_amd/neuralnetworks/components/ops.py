"""Tensor plumbing ops of the hot path (reference: nabu/neuralnetworks/components/ops.py)."""
import weakref

import numpy as np

from nabu_amd import ops as hip
from nabu_amd.autodiff import record, requires_grad, SeqLen


# A-priori magnitude bounds of tensors on the path: id(tensor) -> (weak reference, bound).  layer.blstm records
# |out| <= 1 (o tanh c), the plumbing ops below carry the bound along, and the next layer hands it to the C ABI
# (nabu_blstm_desc.x_bound), whose f16x3 operand packs then take their row scales from it instead of measuring x.
_BOUNDS = {}


# The bound is a promise about the tensor OBJECT: tensors that carry one must not be modified in place (a stale bound makes
# the f16x3 packs overflow fp16 silently).  NABU_CHECK_X_BOUND=1 (layer.CHECK_X_BOUND) measures and asserts at every layer.
def set_value_bound(tensor, bound):
    key = id(tensor)
    _BOUNDS[key] = (weakref.ref(tensor, lambda _r, k=key: _BOUNDS.pop(k, None)), float(bound))


def value_bound(tensor):
    """the recorded bound on |tensor|, 0.0 when nothing is known"""
    e = _BOUNDS.get(id(tensor))
    return e[1] if e is not None and e[0]() is tensor else 0.0


# PACKED COMPANIONS of tensors on the path (include/nabu_hip.h, nabu_blstm_desc ABI version 3): id(tensor) -> (weak reference,
# stack, (rows, cols)) — layer.blstm asks the forward recurrent kernel to write its output ALSO as the next layer's
# f16x3 operands (for the frame stacking `stack` that pyramid_stack is about to apply), and the next layer.blstm hands
# them to the C ABI instead of packing its input.  Only the stacking view carries a companion along; any op that makes a
# new tensor (dropout, noise) drops it, and the next layer packs for itself as before.
_PACKED = {}


def set_packed(tensor, stack, bufs):
    key = id(tensor)
    _PACKED[key] = (weakref.ref(tensor, lambda _r, k=key: _PACKED.pop(k, None)), int(stack), bufs)


def packed(tensor, stack):
    """the (rows, cols) companion of `tensor` written for frame stacking `stack`, or None"""
    e = _PACKED.get(id(tensor))
    return e[2] if e is not None and e[0]() is tensor and e[1] == stack else None


# Zero-filled device buffers that live as long as the process, keyed by their user (a layer's scope and shape): a
# companion is rewritten in full by every forward call and its padding stays zero, so the same buffer serves step after
# step.  `holder` = the Tape whose backward pass still reads the content (None: nobody behind this call): a forward pass
# that finds the buffer held by ANOTHER tape whose backward pass has not run yet (two forward passes of one layer before
# the first backward) gets a buffer of its own instead of overwriting it.
_RESIDENT = {}


def resident_zeros(key, nbytes, device, holder=None):
    import torch
    slots = _RESIDENT.setdefault(key, [])
    for slot in slots:
        buf, ref = slot
        h = ref() if ref is not None else None
        if (h is None or h is holder or not h.ops) and buf.numel() == nbytes and buf.device == device:
            slot[1] = weakref.ref(holder) if holder is not None else None
            return buf
    buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
    slots.append([buf, weakref.ref(holder) if holder is not None else None])
    return buf


def pyramid_stack(inputs, sequence_lengths, numsteps, axis=2, scope=None):
    """Concatenate ``numsteps`` consecutive frames on the feature axis
    (reference ops.py:6-60).

    inputs [B,T,F] batch-major contiguous -> [B, ceil(T/numsteps), numsteps*F];
    lengths -> ceil(len/numsteps) (ops.py:56-58).  On a batch-major buffer the
    stack of consecutive frames IS a reshape, so for T % numsteps == 0 this is a
    free view; otherwise the time axis is zero-padded first (ops.py:32-38)."""
    if axis != 2:
        raise Exception('pyramid_stack: only axis=2 is supported')
    B, T, F = inputs.shape
    Tp = -(-T // numsteps) * numsteps
    src = inputs if Tp == T else hip.pad_time(inputs, Tp)
    outputs = src.view(B, Tp // numsteps, numsteps * F)
    if value_bound(inputs):
        set_value_bound(outputs, value_bound(inputs))
    if Tp == T and packed(inputs, numsteps) is not None:        # the companion was written for exactly this stacking
        set_packed(outputs, 1, packed(inputs, numsteps))

    def backward(dout):
        d = dout.reshape(B, Tp, F)
        return [d if Tp == T else hip.unpad_time(d, T)]
    record([inputs], [outputs], backward)
    lens = SeqLen.wrap(sequence_lengths)
    # (a length vector is immutable: the stacked lengths of a SeqLen OBJECT that comes round again — a batch kept on the
    # device, the layers of one step sharing it — are derived once)
    cache = lens.__dict__.setdefault('_stacked', {})
    if numsteps not in cache:
        new_host = -(-lens.host // numsteps)
        # the device copy is derived on the device: an upload here would block the host on the stream
        cache[numsteps] = SeqLen(new_host.astype(np.int32), dev_tensor=hip.ceil_div_i32(lens.dev, numsteps))
    return outputs, cache[numsteps]


def dense_sequence_to_sparse(sequences, sequence_lengths):
    """The reference converts dense targets to a tf.SparseTensor for tf.nn.ctc_loss
    (ops.py:121-145).  The HIP CTC kernel reads the dense [B,Lmax] labels plus the
    length vector directly, so this returns them unchanged (kept for API parity)."""
    return sequences, sequence_lengths


def seq_dropout(x, keep_prob, rng_state):
    """tf.nn.dropout(x, keep_prob) with a regenerable Philox mask."""
    seed, offset = rng_state.next()
    y = hip.dropout(x, keep_prob, seed, offset)
    if value_bound(x):
        set_value_bound(y, value_bound(x) / keep_prob)

    def backward(dy):
        return [hip.dropout(dy.contiguous(), keep_prob, seed, offset)]
    record([x], [y], backward)
    return y


def input_noise(x, stddev, rng_state):
    """inputs + tf.random_normal(shape, stddev) (listener.py:40-45)."""
    seed, offset = rng_state.next()
    y = hip.gaussian_noise(x, stddev, seed, offset)
    if requires_grad(x):     # (raw features: the noisy copy depends on no parameter either — the first layer then skips
        record([x], [y], lambda dy: [dy])   # its input gradient, two [B T, 8H] x [8H, D] products per step)
    return y


class RngState(object):
    """Seed + running offset for the counter-based device RNG."""

    def __init__(self, seed=0):
        self.seed = int(seed)
        self.offset = 0

    def next(self):
        self.offset += 1
        return self.seed, self.offset


_rng = RngState(0)


def global_rng():
    return _rng


def set_seed(seed):
    _rng.seed = int(seed)
    _rng.offset = 0
