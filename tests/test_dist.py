"""world_size-2 tests of the data-parallel host protocol on CPU (gloo): process
group bring-up, per-rank batch sharding, and the update order of Trainer._update
(clip per replica -> all-reduce(sum) -> Adam with grad_scale 1/world) checked
against the oracle.  The HIP kernels are replaced by NumPy stand-ins here: only
the host logic is under test (the kernels themselves are covered by -m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from oracle import nabu_oracle as O
    from nabu_amd import ops as hip
    from nabu_amd import recipes
    from nabu_amd.computing import dist
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    from nabu_amd.processing.synthetic import SyntheticData

    server = dist.create_server(backend='gloo')
    assert (server.rank, server.world_size) == (rank, world)
    t = torch.full((4,), float(rank + 1))
    server.all_reduce_sum_(t)
    assert torch.all(t == 3.0)
    b = torch.full((3,), float(rank))
    server.broadcast_(b, 0)
    assert torch.all(b == 0.0)

    # NumPy stand-ins for the three kernels the update calls
    def clip_(g, clip=1.0):
        g.clamp_(-clip, clip)
        return g

    def adam(p, g, m, v, lr_t, b1, b2, eps, clip, gscale):
        x = (g * gscale).clamp(-clip, clip)
        m.mul_(b1).add_((1 - b1) * x)
        v.mul_(b2).add_((1 - b2) * x * x)
        p.sub_(lr_t * m / (v.sqrt() + eps))
    hip.clip_, hip.adam_clip_step = clip_, adam

    mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc')
    data = SyntheticData(2, 16, 40, seed=5, batches_per_epoch=10)
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec,
                                             expdir=None, server=server, task_index=rank)
    tr._create_graph()
    assert tr.world == 2
    n = 1000
    rng = np.random.default_rng(7)
    theta0 = rng.normal(size=n)
    grads = [np.random.default_rng(100 + r).normal(0, 2, n) for r in range(world)]   # every rank's gradient
    tr.flat = torch.tensor(theta0.copy())
    tr.flat_grad = torch.tensor(grads[rank].copy())
    tr.adam_m = torch.zeros(n, dtype=torch.float64)
    tr.adam_v = torch.zeros(n, dtype=torch.float64)
    tr._update()
    # oracle: Adam on the MEAN of the per-replica CLIPPED gradients (reference order,
    # trainers/trainer.py:556-569), identical on every rank
    gmean = np.mean([np.clip(g, -1, 1) for g in grads], 0)
    ref, _, _ = O.clip_adam_update(theta0, gmean, np.zeros(n), np.zeros(n), 1, tr.learning_rate())
    np.testing.assert_allclose(tr.flat.numpy(), ref, rtol=1e-12, atol=1e-14)
    # clipping BEFORE averaging matters: the other order gives a different update
    wrong, _, _ = O.clip_adam_update(theta0, np.mean(grads, 0), np.zeros(n), np.zeros(n), 1, tr.learning_rate())
    assert np.abs(wrong - ref).max() > 1e-6
    # each rank reads its own shard of the epoch
    idx = [tr.global_step * world + rank, (tr.global_step + 1) * world + rank]
    q.put((rank, idx, float(tr.flat.sum())))
    server.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_update_matches_oracle():
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]          # disjoint, interleaved batch indices
    assert res[0][2] == res[1][2]                                # replicas stay identical


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from nabu_amd import recipes
    from nabu_amd.computing import dist
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    from nabu_amd.processing.synthetic import SyntheticData
    from tests.bench_standin import FakeModelStep
    server = dist.create_server(backend='gloo')
    finals, logs = {}, {}
    for mode in ('False', 'True'):
        mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **{'trainer.allreduce_buckets': mode})
        tr = trainer_factory.factory('standard')(conf=tc, dataconf=SyntheticData(2, 16, 40), modelconf=mc,
                                                 evaluatorconf=ec, expdir=None, server=server, task_index=rank)
        events = []
        fake = FakeModelStep(tr, server, events)
        # record what the trainer puts on the wire, and when
        orig = server.all_reduce_sum_async

        def spy(t, orig=orig, events=events, tr=tr):
            off = (t.data_ptr() - tr.flat_grad.data_ptr()) // 4
            events.append(('allreduce', int(off), int(t.numel())))
            return orig(t)
        server.all_reduce_sum_async = spy
        for _ in range(3):
            fake.step()
        server.all_reduce_sum_async = orig
        finals[mode] = tr.flat.clone()
        logs[mode] = events
        if mode == 'True':
            assert [b['key'] for b in tr.buckets] == ['Listener/features/layer0', 'Listener/features/layer1',
                                                      'Listener/features/layer2', 'decoder']
            assert tr.buckets[0]['start'] == 0 and tr.buckets[-1]['end'] == tr.flat_grad.numel()
            assert all(a['end'] == b['start'] for a, b in zip(tr.buckets, tr.buckets[1:]))
    # bucketed == flat to the bit (sum of two replicas: the order cannot matter)
    assert torch.equal(finals['False'], finals['True'])
    # schedule of one bucketed step: a bucket goes on the wire between a recurrent kernel and the
    # products that follow it, and only after the products that wrote it; the decoder's bucket first
    ev = logs['True'][:len(logs['True']) // 3]
    kinds = [e[0] + (str(e[1]) if e[0] != 'allreduce' else '') for e in ev]
    assert kinds == ['products3', 'recurrent2', 'allreduce', 'products2', 'recurrent1', 'allreduce', 'products1',
                     'recurrent0', 'allreduce', 'products0', 'allreduce'], kinds
    assert not any(e[0] == 'allreduce' for e in logs['False'])      # flat mode: one blocking all-reduce
    q.put((rank, float(finals['True'].double().sum())))
    server.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_exchange_equals_flat_exchange_bitwise():
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == res[1][1]


def test_single_process_group_is_trivial(monkeypatch):
    from nabu_amd.computing import dist
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('RANK', raising=False)
    g = dist.create_server()
    assert (g.rank, g.world_size) == (0, 1)
    t = torch.ones(3)
    assert g.all_reduce_sum_(t) is t and g.broadcast_(t) is t
    g.barrier()


def test_ranks_share_devices_is_decided_per_node():
    """computing/dist.py: sharing = more ranks ON THIS NODE than visible GPUs (LOCAL_WORLD_SIZE / LOCAL_RANK), not the
    global world size: a 2-node x 8-GPU job (WORLD_SIZE 16) shares nothing and keeps RCCL"""
    from nabu_amd.computing.dist import ranks_share_devices as share
    assert not share(16, 8, {'LOCAL_WORLD_SIZE': '8', 'LOCAL_RANK': '3', 'RANK': '11'})
    assert share(2, 1, {'LOCAL_WORLD_SIZE': '2', 'LOCAL_RANK': '1', 'RANK': '1'})
    assert share(4, 2, {'LOCAL_RANK': '2', 'RANK': '2'})            # no LOCAL_WORLD_SIZE, but a local rank beyond the devices
    assert not share(8, 8, {'LOCAL_WORLD_SIZE': '8', 'LOCAL_RANK': '7', 'RANK': '7'})
    assert share(2, 1, {})                                          # plain environment: a single-node launch is assumed
    assert not share(2, 0, {'LOCAL_WORLD_SIZE': '2'})               # no GPU: nothing to share (gloo on the host)
