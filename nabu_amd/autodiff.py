"""A minimal reverse-mode tape — the stand-in for ``optimizer.compute_gradients``
(nabu/neuralnetworks/trainers/trainer.py:556-558).

Every hot-path op is a HIP kernel with a hand-written gradient kernel; the tape
only remembers which backward closure to call, in reverse order, and routes the
gradient tensors between them.  No arithmetic happens here except the rare
accumulation of a tensor consumed twice (one axpy kernel)."""
import torch

from . import ops as hip_ops


UNIT = object()      # a backward closure's way of saying "this input's gradient is the unit upstream one"


class _Op(object):
    __slots__ = ('inputs', 'outputs', 'backward', 'params')

    def __init__(self, inputs, outputs, backward, params=()):
        self.inputs, self.outputs, self.backward, self.params = inputs, outputs, backward, params


class Tape(object):
    current = None
    current_backward = None        # the tape whose backward() is running (for Tape.defer from inside a closure)

    def __init__(self):
        self.ops = []
        self.produced = set()
        self._prev = None
        # called with a Variable once the last recorded op that declared it (record(..., params=))
        # has run its backward closure: its gradient is final (bucketed gradient exchange)
        self.on_param_ready = None
        self._deferred = None          # during backward(): [(callable, params)] to run after the last closure
        # called after every deferred callable (its params have been reported ready): the data-parallel trainer
        # starts the gradient buckets that became final — behind the last recurrence nothing needs the chip alone
        self.after_deferred = None

    def __enter__(self):
        self._prev = Tape.current
        Tape.current = self
        return self

    def __exit__(self, *exc):
        Tape.current = self._prev
        return False

    def requires_grad(self, tensor):
        """True if ``tensor`` was produced by a recorded op (i.e. it depends on a
        parameter); raw input features do not need an input gradient."""
        return id(tensor) in self.produced

    def defer(self, fn, params=()):
        """Called from inside a backward closure: run ``fn()`` after the LAST closure of this backward pass (work
        nothing downstream waits for — the weight-gradient products of a recurrent layer).  ``params``: Variables
        whose gradient ``fn`` completes; they are reported ready after it ran.  Outside backward(): runs at once."""
        if self._deferred is None:
            fn()
            return
        for v in params:
            if id(v) in self._pending:
                self._pending[id(v)] += 1
        self._deferred.append((fn, tuple(params)))

    def backward(self, root):
        """Run the recorded backward closures from ``root`` (the scalar loss)."""
        grads = {id(root): None}
        seen = {id(root)}
        pending = {}
        self._pending = pending
        self._deferred = []
        outer, Tape.current_backward = Tape.current_backward, self
        try:
            self._run_backward(grads, seen, pending)
        finally:
            Tape.current_backward = outer
            self._deferred = None

    def _run_backward(self, grads, seen, pending):
        if self.on_param_ready is not None:
            for op in self.ops:
                for v in op.params:
                    pending[id(v)] = pending.get(id(v), 0) + 1

        def done(op):
            for v in op.params:
                if id(v) in pending:
                    pending[id(v)] -= 1
                    if pending[id(v)] == 0:
                        self.on_param_ready(v)
        for op in reversed(self.ops):
            if not any(id(o) in seen for o in op.outputs):
                done(op)
                continue
            gouts = [grads.get(id(o)) for o in op.outputs]
            gins = op.backward(*gouts)
            done(op)
            if gins is None:
                continue
            for inp, g in zip(op.inputs, gins):
                if g is None:
                    continue
                key = id(inp)
                if g is UNIT:                  # e.g. a term of a sum of losses: behaves like a root
                    seen.add(key)
                    grads.setdefault(key, None)
                    continue
                if key in seen and grads.get(key) is not None:
                    hip_ops.axpy_(grads[key], g)
                else:
                    grads[key] = g
                    seen.add(key)
        deferred, self._deferred = self._deferred, None
        for fn, params in deferred:
            fn()
            for v in params:
                if id(v) in pending:
                    pending[id(v)] -= 1
                    if pending[id(v)] == 0:
                        self.on_param_ready(v)
            if self.after_deferred is not None:
                self.after_deferred()
        self.ops = []
        self.produced = set()


def record(inputs, outputs, backward, params=()):
    """Register ``backward(*grad_outputs) -> grad_inputs`` on the active tape (no-op
    outside a tape, e.g. in validation).  ``params``: the Variables whose gradient this
    closure writes (optional; lets the tape report when a gradient is final)."""
    tape = Tape.current
    if tape is None:
        return
    tape.ops.append(_Op(list(inputs), list(outputs), backward, tuple(params)))
    for o in outputs:
        tape.produced.add(id(o))


def requires_grad(tensor):
    tape = Tape.current
    return tape is not None and tape.requires_grad(tensor)


class SeqLen(object):
    """Sequence lengths: a host copy (loop bounds, ceil-div) next to the device copy
    the kernels read.  Behaves like the [batch_size] int32 vector of the reference."""

    def __init__(self, host, device=None, dev_tensor=None):
        import numpy as np
        self.host = np.ascontiguousarray(np.asarray(host, dtype=np.int32))
        if dev_tensor is None:
            if device is None:
                device = torch.device('cuda', torch.cuda.current_device())
            dev_tensor = torch.from_numpy(self.host).to(device)
        self.dev = dev_tensor

    @classmethod
    def wrap(cls, x, device=None):
        if isinstance(x, SeqLen):
            return x
        if isinstance(x, torch.Tensor):
            if x.is_cuda:
                return cls(x.detach().cpu().numpy(), dev_tensor=x.to(torch.int32).contiguous())
            return cls(x.numpy(), device)
        return cls(x, device)

    def max(self):
        return int(self.host.max()) if self.host.size else 0

    def __len__(self):
        return len(self.host)

    def numpy(self):
        return self.host

    def __repr__(self):
        return 'SeqLen(%s)' % self.host
