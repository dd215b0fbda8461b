"""Builds libnabu_hip.so (gfx950) in-tree with hipcc.

The shared library is a plain C-ABI object (include/nabu_hip.h): no torch types,
no pybind.  It is built next to this file so that it travels to the GPU box with
the repository snapshot.  hipcc cross-compiles without a GPU."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libnabu_hip.so')
STAMP = os.path.join(HERE, '.libnabu_hip.stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast',
         '-Wall', '-Wno-unused-function']


# per-file additions (lstm_persist_mxh_bwd.hip: its header says why)
EXTRA_FLAGS = {'lstm_persist_mxh_bwd.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest():
    h = hashlib.sha256((' '.join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    files = _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h') or f.endswith('.inc'))
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'nabu_hip.h'))
    for f in files:
        with open(f, 'rb') as fid:
            h.update(f.encode() + b'\0' + fid.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every .hip source for gfx950 and link libnabu_hip.so.  No-op when
    the sources are unchanged."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fid:
            if fid.read().strip() == dig:
                return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as fid:
        fid.write(dig)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
