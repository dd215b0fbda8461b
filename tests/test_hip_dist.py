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
