"""RCCL smoke on ONE GPU (run by tests/test_hip_dist.py in a subprocess): a 1-rank NCCL process group
whose collectives are really issued, driving the data-parallel branch of Trainer._update
(per-replica clip -> all-reduce of the flat gradient buffer -> fused mean + Adam) between the
persistent recurrent kernels of consecutive steps.  Checks that RCCL initialises in this
environment, that stream ordering between the collective and the HIP kernels holds (no time-out of
the persistent kernels' bounded spins), and that the result equals the single-process update."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29517')

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402

from nabu_amd import recipes                   # noqa: E402
from nabu_amd.computing.dist import ProcessGroup                      # noqa: E402
from nabu_amd.neuralnetworks.trainers import trainer_factory, loss_functions   # noqa: E402
from nabu_amd.processing.synthetic import SyntheticData               # noqa: E402


class Forced(ProcessGroup):
    """world_size 1, but every collective is issued (the production class skips them at world 1)"""

    def all_reduce_sum_(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_sum_async(self, t):
        self.async_calls = getattr(self, 'async_calls', 0) + 1
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def broadcast_(self, t, src=0):
        dist.broadcast(t, src)
        return t

    def barrier(self):
        dist.barrier()


def losses(server, force_dp, buckets=False):
    over = {'encoder.num_units': 64, 'trainer.batch_size': 8, 'trainer.allreduce_buckets': str(buckets)}
    mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **over)
    data = SyntheticData(8, 128, 40, min_frames=100, min_labels=2, max_labels=6, time_reduction=8, seed=9)
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec, expdir=None,
                                             server=server, task_index=0)
    tr._create_graph()
    if force_dp:
        tr.world = 1.0000001        # > 1: take the data-parallel branch; the mean over "ranks" stays the sum
    out = [float(tr.step(tr.to_device(data.batch(s))).item()) for s in range(4)]
    loss_functions.check_status()
    return out


def main():
    dist.init_process_group('nccl', rank=0, world_size=1)
    a = losses(Forced(0, 1, 'nccl'), True)
    b = losses(None, False)
    # the bucketed exchange: per-layer buckets started from the hook nabu_blstm_bwd calls between its
    # recurrent kernel and its dense products (a C -> Python callback on the real GPU path), asynchronous
    # collectives on RCCL's stream, joined before the next persistent launch
    srv = Forced(0, 1, 'nccl')
    c = losses(srv, True, buckets=True)
    dist.destroy_process_group()
    assert np.allclose(a, b, rtol=1e-5), (a, b)
    assert np.allclose(c, b, rtol=1e-5), (c, b)
    assert srv.async_calls == 4 * 5, srv.async_calls          # 4 steps x (4 encoder layers + the decoder)
    print('RCCL_SMOKE_OK', a)


if __name__ == '__main__':
    main()
