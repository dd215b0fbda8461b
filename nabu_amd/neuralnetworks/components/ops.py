"""Tensor plumbing ops of the hot path (reference: nabu/neuralnetworks/components/ops.py)."""
import weakref

import numpy as np

from nabu_amd import ops as hip
from nabu_amd.autodiff import record, SeqLen


# A-priori magnitude bounds of tensors on the path: id(tensor) -> (weak reference, bound).  layer.blstm records
# |out| <= 1 (o tanh c), the plumbing ops below carry the bound along, and the next layer hands it to the C ABI
# (nabu_blstm_desc.x_bound), whose f16x3 operand packs then take their row scales from it instead of measuring x.
_BOUNDS = {}


def set_value_bound(tensor, bound):
    key = id(tensor)
    _BOUNDS[key] = (weakref.ref(tensor, lambda _r, k=key: _BOUNDS.pop(k, None)), float(bound))


def value_bound(tensor):
    """the recorded bound on |tensor|, 0.0 when nothing is known"""
    e = _BOUNDS.get(id(tensor))
    return e[1] if e is not None and e[0]() is tensor else 0.0


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
    record([x], [y], lambda dy: [dy])
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
