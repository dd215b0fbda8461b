"""Second tenant on the GPU while the persistent recurrence runs (run by tests/test_hip_dist.py in a subprocess).

The persistent kernels assume that all workgroups of a launch become co-resident; the launch validates the grid
against the occupancy query, but a SECOND tenant — RCCL kernels of a data-parallel exchange, or any other stream —
can hold wave slots / LDS on some CUs when the launch arrives.  This script keeps a side stream busy with real RCCL
collectives (1-rank NCCL group: the all-reduce kernels are launched) and with streaming elementwise kernels, and runs
a cfg2-shaped BLSTM layer (32 x T x 512: 512 workgroups, two per CU) forward + backward on the main stream at the same
time, many times.  Every iteration must either reproduce the solo run bit for bit or fail CLEANLY: the bounded spins
time out, the status word is set and ops.check_persist_status() raises — never a hang, never a wrong number.  After a
clean failure the next solo run must work again."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29519')

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402

from nabu_amd import ops, _hip                 # noqa: E402


def main():
    dist.init_process_group('nccl', rank=0, world_size=1)
    torch.manual_seed(0)
    B, T, D, H = 32, 160, 256, 512
    x = torch.randn(B, T, D, device='cuda') * 0.1
    lens = torch.full((B,), T, dtype=torch.int32, device='cuda')
    p = [torch.randn(s, device='cuda') * 0.03 for s in [(D + H, 4 * H), (4 * H,), (D + H, 4 * H), (4 * H,)]]
    dout = torch.randn(B, T, 2 * H, device='cuda')
    plan = ops.BlstmPlan(B, T, D, H, T, ops.LSTM_PERSISTENT)
    reserve = torch.zeros(plan.reserve_bytes, dtype=torch.uint8, device='cuda')

    def layer():
        out = torch.zeros(B, T, 2 * H, device='cuda')
        g = [torch.zeros_like(q) for q in p]
        dx = torch.zeros_like(x)
        ops.blstm_fwd(plan, x, lens, p[0], p[1], p[2], p[3], out, reserve)
        ops.blstm_bwd(plan, x, lens, p[0], p[2], out, dout, reserve, dx, g[0], g[1], g[2], g[3])
        return [out, dx] + g

    ref = layer()
    torch.cuda.synchronize()
    ops.check_persist_status()
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device='cuda')             # 256 MB
    grads = torch.randn(32 << 20, device='cuda')           # 128 MB "flat gradient"
    ok = failed = 0
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(6):
                dist.all_reduce(grads, op=dist.ReduceOp.SUM)                   # RCCL kernels on its own stream
                big.mul_(1.0000001).add_(1e-9)                                 # streaming tenant
        got = layer()                                                          # persistent launches meet the tenant
        torch.cuda.synchronize()
        try:
            ops.check_persist_status()
        except _hip.NabuHipError as e:
            failed += 1
            assert 'timed out' in str(e), e
            again = layer()
            torch.cuda.synchronize()
            ops.check_persist_status()                                         # recovered
            assert all(torch.equal(a, b) for a, b in zip(again, ref))
            continue
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), 'wrong result beside a second tenant (iteration %d)' % it
        ok += 1
    dist.destroy_process_group()
    print('PERSIST_TENANT_OK ok=%d clean_failures=%d' % (ok, failed))


if __name__ == '__main__':
    main()
