"""The reverse-mode tape's deferral (nabu_amd/autodiff.py, Tape.defer): work nothing downstream waits for — the
weight-gradient products of the recurrent layers — runs after the LAST backward closure, in the order it was deferred;
the gradients it completes are reported final only then, and the data-parallel trainer's hook fires after each."""
import torch

from nabu_amd.autodiff import Tape, record


class Var(object):
    def __init__(self, name):
        self.name = name


def test_deferred_work_runs_last_and_reports_its_params():
    log = []
    k = [Var('k0'), Var('k1')]
    b = [Var('b0'), Var('b1')]
    with Tape() as tape:
        t = torch.zeros(1)
        for i in range(2):
            out = torch.zeros(1)

            def backward(g, i=i):
                log.append('data%d' % i)
                Tape.current_backward.defer(lambda: log.append('weights%d' % i), params=(k[i],))
                return [torch.zeros(1)]
            record([t], [out], backward, params=(k[i], b[i]))
            t = out
    tape.on_param_ready = lambda v: log.append('ready:' + v.name)
    tape.after_deferred = lambda: log.append('hook')
    tape.backward(t)
    # layer 1 (recorded last) runs first; biases are final with the data part, kernels only after the deferred part
    assert log == ['data1', 'ready:b1', 'data0', 'ready:b0', 'weights1', 'ready:k1', 'hook', 'weights0', 'ready:k0', 'hook']
    # outside a backward pass defer() runs at once
    seen = []
    Tape().defer(lambda: seen.append(1))
    assert seen == [1]


def test_deferred_list_is_cleared_after_a_failing_backward():
    with Tape() as tape:
        t = torch.zeros(1)
        out = torch.zeros(1)

        def backward(g):
            Tape.current_backward.defer(lambda: None)
            raise RuntimeError('boom')
        record([t], [out], backward)
    try:
        tape.backward(out)
    except RuntimeError:
        pass
    assert tape._deferred is None and Tape.current_backward is None
