"""Data-parallel process group — replaces nabu/computing/create_server.py
(tf.train.Server over gRPC, parameter servers, ssh tunnels) on one MI355X node.

One process per GPU (launched by torch.distributed.run); gradients are exchanged
with ONE all-reduce of the flat fp32 gradient buffer per training step over
RCCL/xGMI (backend "nccl" is RCCL on ROCm).  ``gloo`` is accepted for the CPU
tests of the host-side protocol."""
import os

import torch
import torch.distributed as dist


class ProcessGroup(object):
    """Handle passed to Trainer(server=...)."""

    def __init__(self, rank, world_size, backend):
        self.rank, self.world_size, self.backend = rank, world_size, backend
        self.shared_devices = False

    def _staged(self, tensor):
        """gloo with device tensors (ranks sharing a GPU: bench.py's first-contact mode, tests): through the host"""
        return self.backend == 'gloo' and tensor.is_cuda

    def all_reduce_sum_(self, tensor):
        if self.world_size > 1:
            if self._staged(tensor):
                host = tensor.detach().cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                tensor.copy_(host)
            else:
                dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        return tensor

    def all_reduce_sum_async(self, tensor):
        """start the all-reduce and return a handle whose wait() joins the caller's stream with
        the collective's (RCCL: stream-ordered, the host does not block; gloo: blocks)"""
        if self.world_size > 1:
            if self._staged(tensor):
                self.all_reduce_sum_(tensor)
                return None
            return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def broadcast_(self, tensor, src=0):
        if self.world_size > 1:
            if self._staged(tensor):
                host = tensor.detach().cpu()
                dist.broadcast(host, src)
                tensor.copy_(host)
            else:
                dist.broadcast(tensor, src)
        return tensor

    def barrier(self):
        if self.world_size > 1:
            dist.barrier()

    def shutdown(self):
        if self.world_size > 1 and dist.is_initialized():
            dist.destroy_process_group()


def ranks_share_devices(world, ngpu, env=None):
    """True when THIS NODE runs more ranks than it has GPUs, i.e. ranks share a device.  Decided per node: the launcher's
    LOCAL_WORLD_SIZE (torch.distributed.run sets it) against the visible GPU count, or a LOCAL_RANK beyond the last
    device; the global WORLD_SIZE says nothing about it on a multi-node job (2 nodes x 8 GPUs: WORLD_SIZE 16, 8 per
    node, nothing shared — such a job keeps RCCL).  Without either variable a single-node launch is assumed."""
    env = os.environ if env is None else env
    if ngpu <= 0:
        return False
    if 'LOCAL_WORLD_SIZE' in env:
        return int(env['LOCAL_WORLD_SIZE']) > ngpu
    if 'LOCAL_RANK' in env and int(env['LOCAL_RANK']) >= ngpu:
        return True
    return world > ngpu


def create_server(backend=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
    MASTER_PORT (set by torch.distributed.run).  Single-process when unset —
    the analogue of create_local_server() (create_server.py:23-25)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return ProcessGroup(0, 1, None)
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared = ranks_share_devices(world, ngpu)
    if backend is None:
        backend = 'gloo' if (shared or not ngpu) else 'nccl'
    if ngpu:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % ngpu)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    pg = ProcessGroup(rank, world, backend)
    pg.shared_devices = shared
    return pg
