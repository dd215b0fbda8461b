"""Variables and scopes — the stand-in for TF's variable store on the hot path.

The reference creates its parameters lazily under nested ``tf.variable_scope``s
with ``tf.AUTO_REUSE`` (ed_encoder.py:35-38) and finds them again by scope
prefix (ed_encoder.py:84-96).  Here a ``VariableStore`` keeps named fp32 device
tensors with the same names; ``flatten()`` re-homes every trainable variable in
one flat parameter buffer (plus a flat gradient buffer) so that the optimiser is
one fused kernel and the data-parallel exchange is one RCCL all-reduce."""
import contextlib
import threading

import numpy as np
import torch

_state = threading.local()


def _scopes():
    if not hasattr(_state, 'scopes'):
        _state.scopes = []
    return _state.scopes


@contextlib.contextmanager
def variable_scope(name):
    """Nested name scope: variables created inside are called ``a/b/name``."""
    _scopes().append(name)
    try:
        yield '/'.join(_scopes())
    finally:
        _scopes().pop()


def current_scope():
    return '/'.join(_scopes())


class Variable(object):
    """A named parameter: ``data`` and ``grad`` are device tensors (views into the
    flat buffers once the store is flattened)."""

    def __init__(self, name, data, trainable=True):
        self.name = name
        self.data = data
        self.grad = None
        self.trainable = trainable

    @property
    def shape(self):
        return tuple(self.data.shape)

    def numel(self):
        return self.data.numel()

    def __repr__(self):
        return 'Variable(%s, %s)' % (self.name, self.shape)


def glorot_uniform(rng, shape):
    """TF-1.8 scope default initialiser (glorot_uniform_initializer); for rank-1
    shapes fan_in = fan_out = shape[0] (this is what LayerNormBasicLSTMCell's
    bias gets: SURVEY.md 8(a) A5)."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def zeros(rng, shape):
    return np.zeros(shape, np.float32)


class VariableStore(object):
    """Ordered name -> Variable map with AUTO_REUSE semantics."""

    def __init__(self, seed=0, device=None):
        self.vars = {}
        self.order = []
        self.rng = np.random.default_rng(seed)
        self.device = device
        self.flat = None
        self.flat_grad = None
        self.restore = {}          # name -> array: values variables take when they are created

    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError('nabu_amd needs a GPU: parameters live in HBM and there is '
                                   'no CPU fallback')
            self.device = torch.device('cuda', torch.cuda.current_device())
        return self.device

    def get_variable(self, name, shape, initializer=glorot_uniform, trainable=True):
        full = current_scope() + '/' + name if current_scope() else name
        var = self.vars.get(full)
        if var is not None:
            if var.shape != tuple(shape):
                raise Exception('variable %s exists with shape %s, requested %s'
                                % (full, var.shape, tuple(shape)))
            return var
        if self.flat is not None:
            raise Exception('variable %s created after the store was flattened' % full)
        value = initializer(self.rng, tuple(shape))
        if full in self.restore:                        # a loaded checkpoint wins over the initialiser
            value = np.asarray(self.restore.pop(full), np.float32).reshape(tuple(shape))
        var = Variable(full, torch.from_numpy(np.ascontiguousarray(value)).to(self._dev()), trainable)
        self.vars[full] = var
        self.order.append(full)
        return var

    def variables(self, prefix=''):
        """Variables whose name starts with ``prefix`` (tf.get_collection(scope=...))."""
        return [self.vars[n] for n in self.order if n.startswith(prefix)]

    def trainable_variables(self):
        return [v for v in self.variables() if v.trainable]

    def num_params(self):
        return sum(v.numel() for v in self.trainable_variables())

    def flatten(self):
        """Move every trainable variable into one flat fp32 buffer (each variable
        starts on a 16-byte boundary) and allocate the flat gradient buffer."""
        if self.flat is not None:
            return self.flat, self.flat_grad
        tv = self.trainable_variables()
        offs, total = [], 0
        for v in tv:
            offs.append(total)
            total += (v.numel() + 3) // 4 * 4
        dev = self._dev()
        self.flat = torch.zeros(max(total, 4), dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros_like(self.flat)
        for v, o in zip(tv, offs):
            n = v.numel()
            self.flat[o:o + n].copy_(v.data.reshape(-1))
            v.data = self.flat[o:o + n].view(v.shape)
            v.grad = self.flat_grad[o:o + n].view(v.shape)
            v.offset = o
        return self.flat, self.flat_grad

    # -- checkpointing (plain dict of host arrays keyed by TF-style names) ----
    def state_dict(self):
        return {n: self.vars[n].data.detach().cpu().numpy() for n in self.order}

    def load_state_dict(self, state):
        for n, value in state.items():
            if n not in self.vars:
                raise Exception('unknown variable %s' % n)
            self.vars[n].data.copy_(torch.from_numpy(np.ascontiguousarray(value, np.float32)))

    def restore_from(self, state):
        '''LoadAtBegin (components/hooks.py:6-28) for a store whose variables are created on first
        use: existing variables are overwritten now, the others when they come into existence'''
        for n, value in state.items():
            if n in self.vars:
                self.vars[n].data.copy_(torch.from_numpy(np.ascontiguousarray(value, np.float32)))
            else:
                self.restore[n] = np.asarray(value)


_default = [None]


def default_store():
    if _default[0] is None:
        _default[0] = VariableStore()
    return _default[0]


@contextlib.contextmanager
def as_default(store):
    old = _default[0]
    _default[0] = store
    try:
        yield store
    finally:
        _default[0] = old


def get_variable(name, shape, initializer=glorot_uniform, trainable=True):
    return default_store().get_variable(name, shape, initializer, trainable)
