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


def _two_rank_line(extra, with_cache=True):
    """python bench.py --gpus 2 --shrink ...: two REAL ranks on the one-GPU test box (they share the device, so the
    group is gloo and the recurrence runs step-wise), 2 timed training steps, the whole JSON line"""
    import json
    import socket
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = os.path.join(root, 'gpurun_out', 'cpu_baseline_cache.json')
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    keep = open(cache).read() if os.path.exists(cache) else None
    if with_cache:
        with open(cache, 'w') as fid:      # what an N = 1 run on this host leaves behind; the N > 1 line carries it
            json.dump({'host': socket.gethostname(), 'time': time.time(),
                       'cpu_baseline': {'value': 0.5, 'unit': 'utterances/sec', 'cores': 8, 'kind': 'port', 'sample': 'test'}}, fid)
    elif os.path.exists(cache):
        os.remove(cache)
    try:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--shrink', '--steps', '2',
                            '--warmup', '1', '--no-alt', '--no-gemm-roofline'] + extra, env=env, capture_output=True,
                           text=True, timeout=900)
    finally:
        if keep is None:
            if os.path.exists(cache):
                os.remove(cache)
        else:
            with open(cache, 'w') as fid:
                fid.write(keep)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])


def test_bench_two_rank_line_end_to_end_on_one_gpu():
    """the N > 1 JSON line, every object of it, from two real ranks and two real training steps (flat exchange), then
    the bucketed exchange: same loss to the bit, buckets on the wire in backward order from the phase hooks"""
    flat = _two_rank_line([])
    assert flat['n_gpus'] == 2 and flat['ranks']['world_size_seen'] == 2 and flat['ranks']['backend'] == 'gloo'
    assert flat['ranks']['ranks_share_devices'] and flat['config']['shrunk'] and flat['config']['recurrent_path'] == 'stepwise'
    assert flat['config']['global_batch'] == 8 and flat['config']['parallelism'] == 'dp2'
    assert len(flat['ranks']['ms_per_step_per_rank']) == 2 and all(v > 0 for v in flat['ranks']['ms_per_step_per_rank'])
    assert flat['ranks']['allreduce'] == 'flat' and all(v > 0 for v in flat['ranks']['allreduce_ms_per_step'])
    assert flat['cpu_baseline']['value'] == 0.5 and 'carried_from' in flat['cpu_baseline']
    assert flat['roofline']['bound'] == 'hbm' and flat['roofline']['frac'] > 0 and flat['value'] > 0
    assert flat['scaling'] == 'weak' and flat['higher_is_better'] is True
    # which path every rank ran is part of the line (here: ranks share the device -> step-wise recurrence, no decoder)
    assert flat['ranks']['recurrence_persistent_per_rank'] == [0, 0] and flat['ranks']['decoder_persistent_per_rank'] == [-1, -1]
    # the replicas hold bit-identical weights before the timed region and after its last step (checksum all-gather: bench.py
    # raises on every rank otherwise), and the line says which collective library carried the exchange
    rw = flat['ranks']['replica_weights']
    assert rw['before_timed_region']['identical'] and rw['after_timed_region']['identical']
    assert rw['before_timed_region']['checksum'] != rw['after_timed_region']['checksum']        # (training moved them)
    lib = flat['ranks']['collective_library']
    assert lib['backend'] == 'gloo' and 'nccl_algo_set' in lib and 'env' in lib and 'io_links_seen' in lib
    # three repeats of the K timed steps in the same command, the headline being the first
    assert len(flat['ms_per_step_repeats']) == 3 and flat['ms_per_step_repeats'][0] == flat['ms_per_step']
    buck = _two_rank_line(['--allreduce', 'bucketed'])
    assert buck['ranks']['allreduce'] == 'bucketed'
    assert buck['final_loss'] == flat['final_loss']
    sched = buck['ranks']['bucket_schedule_last_step']
    assert [k for k, _ in sched] == ['decoder', 'Listener/features/layer3', 'Listener/features/layer2',
                                     'Listener/features/layer1', 'Listener/features/layer0'], sched
    # every bucket but the last one to become final leaves from a hook between kernels, not at the optimiser
    assert [w for _, w in sched][:4] == ['hook'] * 4, sched


def test_bench_two_rank_line_without_an_n1_run_and_both_exchanges():
    """first contact with a multi-GPU node: no N = 1 run of this host left a cache, and still the line carries a
    cpu_baseline (the last committed N = 1 figure, labelled with where it came from); --allreduce both times the flat and
    the bucketed exchange in ONE run"""
    line = _two_rank_line(['--allreduce', 'both'], with_cache=False)
    cb = line['cpu_baseline']
    assert cb is not None and cb['value'] > 0 and cb['kind'] == 'port'
    assert 'profiles/cpu_baseline_last.json' in cb['carried_from']
    both = line['ranks']['allreduce_both']
    assert line['ranks']['allreduce'] == 'flat' and both['flat']['ms_per_step'] == line['ms_per_step']
    assert both['bucketed']['ms_per_step'] > 0 and len(both['bucketed']['exposed_allreduce_ms_per_step']) == 2
    assert [k for k, _ in both['bucketed']['bucket_schedule_last_step']][0] == 'decoder'


def test_bench_cfg3_line_carries_the_decoder_roofline():
    """python bench.py --workload cfg3: the line of a recipe with a Speller prices the decoder's two calls against
    their own algorithmic bytes (roofline_decoder: keys + values per decoder step and pass over the calls' wall time)"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'cfg3', '--steps', '2', '--warmup', '1',
                        '--repeats', '1', '--no-cpu-baseline', '--no-alt'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    rd = line['roofline_decoder']
    assert rd['bound'] == 'hbm' and rd['calls_timed'] == 2 and rd['utterances'] == 32 and rd['frames'] == 125
    assert rd['bytes_per_decoder_step_and_pass'] == 4 * 32 * 125 * (512 + 1024)
    for k in ('fwd', 'bwd'):
        assert 0.0 < rd[k]['frac'] < 1.0 and rd[k]['ms_per_call'] > 0.1
    assert abs(rd['ms_per_step'] - rd['fwd']['ms_per_call'] - rd['bwd']['ms_per_call']) < 0.01
    assert line['ranks']['decoder_persistent_per_rank'] == [3]
