"""Build-time disassembly check of the ONE matrix instruction the compiler cannot see: lstm_persist_mxf.hip issues
the l plane of W_h through inline assembly with the weights in ACCUMULATION registers (`mxf_mfma_acc`, lstm_persist_mxh.h).
hipcc's hazard recognizer does not look inside inline assembly, so nothing pads a vector-ALU read (or a v_accvgpr_read)
of the accumulator behind it.  The pattern is only safe because the next instruction that touches that accumulator is the
compiler's own matrix instruction of the h plane (whose hazards hipcc does pad).  This test compiles the file for gfx950
(no GPU needed) and asserts exactly that for every such instruction in every kernel — a compiler upgrade or a change of
register pressure that breaks the pattern fails here, not as silently wrong recurrent sums."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(tok):
    m = re.match(r'([av])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'([av])(\d+)$', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def test_inline_asm_mfma_is_always_followed_by_a_compiler_mfma_on_the_same_accumulators(tmp_path):
    from nabu_amd import build
    src = os.path.join(ROOT, 'nabu_amd', 'csrc', 'lstm_persist_mxf.hip')
    out = str(tmp_path / 'mxf.s')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    cmd = [hipcc] + build.FLAGS + build.EXTRA_FLAGS.get('lstm_persist_mxf.hip', []) + ['--cuda-device-only', '-S', src, '-o', out]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    lines = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((';', '.', '//'))]
    found = 0
    for i, l in enumerate(lines):
        if not l.startswith('v_mfma_f32_16x16x32_f16'):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        if not ops[1].startswith('a'):            # the compiler keeps matrix A operands in ordinary registers: A in a[..] = the asm
            continue
        found += 1
        acc = _regs(ops[0])
        assert acc and _regs(ops[3]) == acc, l     # accumulates in place
        for nxt in lines[i + 1:i + 400]:
            if nxt.endswith(':') or nxt.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_barrier')):
                pytest.fail('control flow between the asm matrix instruction and the next use of its accumulator: %s ... %s' % (l, nxt))
            toks = re.findall(r'[av]\[\d+:\d+\]|[av]\d+', nxt)
            if any(_regs(t) & acc for t in toks):
                assert nxt.startswith('v_mfma'), 'accumulator of an inline-asm matrix instruction touched by a non-matrix instruction: %s -> %s' % (l, nxt)
                break
        else:
            pytest.fail('no later use of the accumulator found: %s' % l)
    assert found >= 16, 'expected the l-plane instructions of lstm_mxf_fwd/bwd_kernel in the disassembly (found %d)' % found
