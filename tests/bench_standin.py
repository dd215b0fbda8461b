"""Stand-in workload for the CPU test of bench.py's multi-rank protocol (tests/test_bench_launch.py).

bench.py's own main() runs unchanged — argument parsing, rank discovery from the environment that
torch.distributed.run prepares, warm-up, barrier + sync bracketing, MAX over ranks, the JSON line of
rank 0 — with two substitutions made here, in tests/: the process group is gloo (no GPU in the build
container) and the workload is a tiny NumPy "training step" that goes through the REAL
Trainer._update (clip -> all-reduce -> Adam) with NumPy stand-ins for the three HIP kernels it calls.
Nothing in the product imports this file."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                                        # noqa: E402
from nabu_amd import ops as hip, recipes                            # noqa: E402
from nabu_amd.computing import dist                                 # noqa: E402
from nabu_amd.neuralnetworks.trainers import trainer_factory        # noqa: E402
from nabu_amd.processing.synthetic import SyntheticData             # noqa: E402


def clip_(g, clip=1.0):
    g.clamp_(-clip, clip)
    return g


def adam(p, g, m, v, lr_t, b1, b2, eps, clip, gscale):
    x = (g * gscale).clamp(-clip, clip)
    m.mul_(b1).add_((1 - b1) * x)
    v.mul_(b2).add_((1 - b2) * x * x)
    p.sub_(lr_t * m / (v.sqrt() + eps))


LAYER_VARS = [('Listener/features/layer%d/BLSTM/bidirectional_rnn/%s/layer_norm_basic_lstm_cell/%s' % (l, d, k), shp)
              for l in range(3) for d in ('fw', 'bw') for k, shp in (('kernel', (24, 16)), ('bias', (16,)))]
DECODER_VARS = [('DNNDecoder/text/outlayer/weights', (8, 5)), ('DNNDecoder/text/outlayer/biases', (5,))]


class FakeModelStep(object):
    """A 'model' of three encoder layers + an output layer whose backward closures behave like the
    real ones towards the trainer: they declare their parameters on the tape, join the communication
    stream before their 'recurrent kernel' (ops.BEFORE_RECURRENT), call the phase hook between the
    'recurrent kernel' and the 'products' (what nabu_blstm_bwd does), and then write their gradients
    into the flat gradient buffer.  Gradients are a function of (rank, step) only."""

    def __init__(self, tr, server, events=None):
        from nabu_amd import variables as vs
        self.tr, self.server, self.events = tr, server, events if events is not None else []
        store = tr.model.store
        store.device = torch.device('cpu')
        with vs.as_default(store):
            for name, shape in LAYER_VARS + DECODER_VARS:
                vs.get_variable(name, list(shape))
        self.hook = [None]
        hip.set_phase_hook = lambda fn: self.hook.__setitem__(0, fn)
        hip.clip_, hip.adam_clip_step = clip_, adam
        tr._create_graph()
        tr._init_optimizer()
        self.k = 0

    def step(self):
        from nabu_amd.autodiff import Tape, record
        self.k += 1
        store = self.tr.model.store
        rng = np.random.default_rng([self.server.rank, self.k])
        groups = [[store.vars[n] for n, _ in LAYER_VARS if '/layer%d/' % l in n] for l in range(3)]
        groups.append([store.vars[n] for n, _ in DECODER_VARS])
        with Tape() as tape:
            t = torch.zeros(1)
            for gi, vars_ in enumerate(groups):
                out = torch.zeros(1)

                def backward(g, vars_=vars_, gi=gi):
                    if gi < 3:                                    # an encoder layer: recurrent kernel first
                        if hip.BEFORE_RECURRENT[0] is not None:
                            hip.BEFORE_RECURRENT[0]()
                        self.events.append(('recurrent', gi))
                        if self.hook[0] is not None:
                            self.hook[0]()
                    for v in vars_:
                        v.grad.copy_(torch.from_numpy(rng.normal(0, 2, tuple(v.shape)).astype(np.float32)))
                    self.events.append(('products', gi))
                    return [torch.zeros(1)]
                record([t], [out], backward, params=vars_)
                t = out
        self.tr._backward_and_update(tape, t)


class StandIn(object):
    units_per_step = 32

    def __init__(self, args, server):
        self.args, self.server = args, server
        over = {'trainer.allreduce_buckets': 'True' if args.allreduce == 'bucketed' else 'False'}
        mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **over)
        self.tr = trainer_factory.factory('standard')(conf=tc, dataconf=SyntheticData(2, 16, 40), modelconf=mc,
                                                      evaluatorconf=ec, expdir=None, server=server,
                                                      task_index=server.rank)
        self.fake = FakeModelStep(self.tr, server)
        assert (self.tr.buckets is not None) == (args.allreduce == 'bucketed' and server.world_size > 1)

    def step(self, i):
        time.sleep(0.002 * (1 + self.server.rank))        # ranks of different speed: the MAX must win
        self.fake.step()

    def sync(self):
        pass

    def check(self):
        pass

    def start_timed_region(self):
        pass

    def end_timed_region(self):
        self.checksums = self.gather(float(self.tr.flat.double().sum()))     # collective: every rank

    def reduce_max(self, values):
        t = torch.tensor(values, dtype=torch.float64)
        if self.server.world_size > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return t.tolist()

    def gather(self, value):
        t = torch.tensor([value], dtype=torch.float64)
        if self.server.world_size == 1:
            return [float(value)]
        out = [torch.zeros_like(t) for _ in range(self.server.world_size)]
        torch.distributed.all_gather(out, t)
        return [float(o.item()) for o in out]

    def allreduce_ms_per_step(self):
        return None

    def bucket_schedule(self):
        return None

    def alt(self, steps):
        return []

    def describe(self, dt):
        return {'metric': 'stand-in', 'dtype': 'f32', 'config': {'workload': 'stand-in'}, 'roofline': None,
                'replica_checksum': self.checksums}

    def wants_cpu_baseline(self):
        return False


if __name__ == '__main__':
    bench.make_server = lambda: dist.create_server(backend='gloo')
    bench.make_workload = StandIn
    bench.main()
