"""GPU: the data-parallel update through a real RCCL process group (1 rank on the one GPU of the
test box; the N > 1 protocol itself is covered by the world_size-2 gloo tests of tests/test_dist.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_all_reduce_between_persistent_kernels():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(here, 'rccl_smoke.py')], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_SMOKE_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_persistent_recurrence_beside_a_second_tenant():
    """RCCL collectives and streaming kernels on a side stream WHILE persistent recurrent launches are in flight:
    bit-identical results or the clean status-word error (tests/persist_tenant.py)"""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(here, 'persist_tenant.py')], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'PERSIST_TENANT_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_two_rank_dry_run_on_one_gpu():
    """bench.py --gpus 2 --dry-run on the one-GPU test box: two ranks are started and rendezvous exactly as a
    measurement would (the ranks share the device, so the group is gloo), one collective runs, no workload"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['ok'] and out['world_size_seen'] == 2 and out['all_reduce_sum'] == 3.0
