"""bench.py's multi-rank path on CPU: `python bench.py --gpus N` with no ranks in the environment
starts its own N ranks (torch.distributed.run over 127.0.0.1) and rank 0 prints ONE JSON line; the
driver's way (torch.distributed.run ... bench.py --gpus N) keeps working.  The HIP workload is
replaced by tests/bench_standin.py (gloo + NumPy stand-ins); everything else is bench.py's own code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, 'tests', 'bench_standin.py')
sys.path.insert(0, ROOT)

import bench                                         # noqa: E402


def json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith('{') and line.endswith('}'):
            out.append(json.loads(line))
    return out


def clean_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['OMP_NUM_THREADS'] = '1'
    return env


def check_line(o, n, steps, warmup):
    assert (o['n_gpus'], o['steps'], o['warmup'], o['scaling'], o['unit']) == (n, steps, warmup, 'weak', 'utterances/sec')
    assert o['ranks']['world_size_seen'] == n and len(o['ranks']['ms_per_step_per_rank']) == n
    assert o['config']['global_batch'] == 32 * n and o['config']['parallelism'] == 'dp%d' % n
    # whole-job value from the MAX over ranks: the slowest rank sleeps n*2 ms per step
    assert o['ms_per_step'] >= max(o['ranks']['ms_per_step_per_rank']) - 1e-3
    assert o['ms_per_step'] >= 2.0 * n
    assert abs(o['value'] - 32 * n / (o['ms_per_step'] * 1e-3)) / o['value'] < 1e-3
    # replicas stay identical under clip -> all-reduce -> Adam
    assert len(set(o['replica_checksum'])) == 1


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks(monkeypatch):
    """no WORLD_SIZE in the environment + --gpus 2: main() must self-launch"""
    calls = {}

    def fake_launch(argv, nproc, script=None):
        calls['argv'], calls['n'] = list(argv), nproc
        return 0
    monkeypatch.setattr(bench, 'self_launch', fake_launch)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(['--gpus', '2', '--steps', '3', '--warmup', '1'])
    assert e.value.code == 0 and calls == {'argv': ['--gpus', '2', '--steps', '3', '--warmup', '1'], 'n': 2}
    cmd = bench.launch_command(['--gpus', '2'], 2, port=1234)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and '127.0.0.1' in cmd
    assert cmd[-3:] == [os.path.join(ROOT, 'bench.py'), '--gpus', '2']


@pytest.mark.timeout(600)
@pytest.mark.parametrize('allreduce', ['flat', 'bucketed'])
def test_self_launched_two_rank_run_prints_one_json_line(allreduce):
    code = ('import sys; sys.path.insert(0, %r); import bench; '
            'sys.exit(bench.self_launch(["--gpus", "2", "--steps", "4", "--warmup", "1", "--allreduce", %r], 2, script=%r))'
            % (ROOT, allreduce, STANDIN))
    r = subprocess.run([sys.executable, '-c', code], env=clean_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    check_line(lines[0], 2, 4, 1)
    assert lines[0]['ranks']['backend'] == 'gloo' and lines[0]['ranks']['allreduce'] == allreduce


@pytest.mark.timeout(600)
def test_driver_style_launch_and_single_rank():
    """the driver's command line (ranks given by torch.distributed.run) and the N=1 default"""
    cmd = bench.launch_command(['--gpus', '2', '--steps', '3', '--warmup', '2'], 2, script=STANDIN)
    r = subprocess.run(cmd, env=clean_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1
    check_line(lines[0], 2, 3, 2)
    r = subprocess.run([sys.executable, STANDIN, '--steps', '2', '--warmup', '1'], env=clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1
    check_line(lines[0], 1, 2, 1)
    # a rank count that contradicts --gpus is an error, not a silent mismatch
    env = clean_env()
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(bench.free_port()))
    r = subprocess.run([sys.executable, '-c',
                        'import sys; sys.path.insert(0, %r); import bench; bench.make_server = lambda: '
                        'type("S", (), {"world_size": 2, "rank": 0})(); bench.main(["--gpus", "4"])' % ROOT],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stderr + r.stdout)


def test_dry_run_starts_two_ranks_and_runs_a_collective():
    """`python bench.py --gpus 2 --dry-run`: self-launch, rendezvous on 127.0.0.1, process group, one all-reduce,
    one JSON line — without the workload (on this CPU box: gloo; the GPU twin is tests/test_hip_dist.py)"""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['ok'] and out['world_size_seen'] == 2 and out['all_reduce_sum'] == 3.0
