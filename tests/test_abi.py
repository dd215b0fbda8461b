"""CPU tests of the drop-in boundary: libnabu_hip.so loads and exports every
symbol that include/nabu_hip.h declares, with ctypes signatures for each."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'nabu_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nabu_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from nabu_amd import build, _hip
    build.build(verbose=False)
    lib = _hip.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), 'libnabu_hip.so does not export %s' % n
        assert n in _hip.SIGNATURES, 'no ctypes signature for %s' % n
    for n in _hip.SIGNATURES:
        assert n in names, '%s bound in _hip.py but not declared in nabu_hip.h' % n
    assert lib.nabu_version() == 3 == _hip.ABI_VERSION


def test_graft_entry_build_checks_the_same_version():
    """__graft_entry__.build() is the driver's "does it build" check: its version assertion must follow the header
    (round 5 bumped NABU_ABI_VERSION and the literal there stayed behind)."""
    from nabu_amd import _hip
    hdr = open(os.path.join(ROOT, 'include', 'nabu_hip.h')).read()
    version = int(re.search(r'#define\s+NABU_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert version == _hip.ABI_VERSION
    src = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    assert '_hip.ABI_VERSION' in src and not re.search(r'nabu_version\(\)\s*==\s*\d', src)


def test_host_side_queries_and_argument_errors():
    """No GPU needed: size queries and argument validation run on the host."""
    import ctypes
    from nabu_amd import _hip
    lib = _hip.lib()
    d = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 32, 1000, 40, 512, 1000, 0, 0)
    res = lib.nabu_blstm_reserve_bytes(ctypes.byref(d))
    assert res == (2 * 32 * 1000 * 2048 + 2 * 32 * 1000 * 512) * 4
    assert lib.nabu_blstm_ws_bytes(ctypes.byref(d)) > 0
    bad = _hip.BlstmDesc(4, 32, 1000, 40, 512, 1000, 0, 0)
    assert lib.nabu_blstm_reserve_bytes(ctypes.byref(bad)) == 0
    odd = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 2, 3, 4, 6, 3, 0, 0)       # H % 4 != 0
    assert lib.nabu_blstm_reserve_bytes(ctypes.byref(odd)) == 0
    assert b'multiple of 4' in lib.nabu_last_error()
    assert lib.nabu_ctc_ws_bytes(32, 125, 60) == 32 * 125 * 121 * 4
    # null pointers are rejected before any launch
    assert lib.nabu_gemm_f32(0, 0, 4, 4, 4, 1.0, None, 4, None, 4, 0.0, None, 4, None, 0, 0, 0,
                             None, 0, None) == -1


def test_blstm_descriptor_versions_and_forward_only_reserve():
    """ABI version 2 of nabu_blstm_desc (x_bound, flags, recurrent_precision): the 32-byte version-1 layout is still
    accepted; a forward-only descriptor's reserve ends behind the activations (no packed dZ^T region); bad values are
    argument errors.  No GPU needed."""
    import ctypes
    from nabu_amd import _hip
    lib = _hip.lib()
    act = (2 * 32 * 500 * 2048 + 2 * 32 * 500 * 512) * 4
    full = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 32, 500, 2048, 512, 500, 0, _hip.GEMM_PRECISIONS['bf16x6'])
    v1 = _hip.BlstmDesc(32, 32, 500, 2048, 512, 500, 0, _hip.GEMM_PRECISIONS['bf16x6'])
    r_full, r_v1 = lib.nabu_blstm_reserve_bytes(ctypes.byref(full)), lib.nabu_blstm_reserve_bytes(ctypes.byref(v1))
    assert r_full == r_v1 >= act
    fwd = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 32, 500, 2048, 512, 500, 0, _hip.GEMM_PRECISIONS['bf16x6'], 0.0,
                         _hip.BLSTM_FWD_ONLY, 0)
    assert lib.nabu_blstm_reserve_bytes(ctypes.byref(fwd)) == act
    for field, value in (('x_bound', -1.0), ('x_bound', float('inf')), ('flags', 2), ('recurrent_precision', 7)):
        bad = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 32, 500, 2048, 512, 500, 0, 0)
        setattr(bad, field, value)
        assert lib.nabu_blstm_reserve_bytes(ctypes.byref(bad)) == 0, field
    exact = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 32, 500, 2048, 512, 500, 0, 0, 0.0, 0, _hip.REC_PRECISIONS['f32'])
    assert lib.nabu_blstm_ws_bytes(ctypes.byref(exact)) > 0


def test_backward_rejects_a_reserve_no_forward_call_wrote():
    """nabu_blstm_bwd{,_data,_weights} compare the layout they derive with the fingerprint nabu_blstm_fwd recorded for
    the reserve (host-side, before any device work): a reserve nobody wrote, or a forward-only descriptor, is
    NABU_EINVAL.  (The matching case — forward then backward — is every GPU test; a layout that changed between the
    passes is tests/test_hip_ops.py::test_backward_rejects_a_reserve_of_another_layout.)"""
    import ctypes
    from nabu_amd import _hip
    lib = _hip.lib()
    d = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 4, 16, 8, 64, 16, 1, 0)
    fake = ctypes.c_void_p(0x1000)           # never dereferenced: the check comes first
    args = dict(x=fake, len=fake, k=fake, out=fake, dout=fake, reserve=ctypes.c_void_p(0x7000), ws=fake)
    rc = lib.nabu_blstm_bwd(ctypes.byref(d), args['x'], args['len'], args['k'], args['k'], args['out'], args['dout'],
                            args['reserve'], None, fake, fake, fake, fake, args['ws'], 1 << 40, None)
    assert rc == -1 and b'no nabu_blstm_fwd call' in lib.nabu_last_error()
    rc = lib.nabu_blstm_bwd_data(ctypes.byref(d), args['x'], args['len'], args['k'], args['k'], args['out'], args['dout'],
                                 args['reserve'], None, fake, fake, args['ws'], 1 << 40, None)
    assert rc == -1 and b'no nabu_blstm_fwd call' in lib.nabu_last_error()
    rc = lib.nabu_blstm_bwd_weights(ctypes.byref(d), args['x'], args['len'], args['out'], args['reserve'], fake, fake,
                                    args['ws'], 1 << 40, None)
    assert rc == -1 and b'no nabu_blstm_fwd call' in lib.nabu_last_error()
    fwd_only = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), 4, 16, 8, 64, 16, 1, 0, 0.0, _hip.BLSTM_FWD_ONLY, 0)
    rc = lib.nabu_blstm_bwd(ctypes.byref(fwd_only), args['x'], args['len'], args['k'], args['k'], args['out'], args['dout'],
                            args['reserve'], None, fake, fake, fake, fake, args['ws'], 1 << 40, None)
    assert rc == -1 and b'NABU_BLSTM_FWD_ONLY' in lib.nabu_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    from nabu_amd import _hip
    with pytest.raises(_hip.NabuHipError):
        _hip.ptr(torch.zeros(4))
    # the package never imports the oracle
    pkg = os.path.join(ROOT, 'nabu_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), f


def test_decoder_entry_points_validate_on_the_host():
    """size queries and argument errors of the inference entry points (decode.hip), no GPU"""
    import ctypes
    from nabu_amd import _hip
    lib = _hip.lib()
    # prefix trees: (1 + T*W) nodes x (parent, label, slot, C-1 children) int32 per utterance
    assert lib.nabu_ctc_beam_ws_bytes(32, 125, 40, 100) == 32 * (1 + 125 * 100) * (3 + 39) * 4
    assert lib.nabu_ctc_beam_ws_bytes(0, 125, 40, 100) == 0
    one = ctypes.c_void_p(16)          # any non-null pointer: validation happens before it is touched
    assert lib.nabu_ctc_beam_search(2, 5, 4, 300, 1, one, one, one, one, None, one, 1 << 30, None) == -2
    assert b'beam_width' in lib.nabu_last_error()
    assert lib.nabu_ctc_beam_search(2, 5, 4, 8, 1, one, one, one, one, None, one, 16, None) == -3    # workspace
    assert lib.nabu_ctc_beam_search(2, 5, 4, 8, 1, None, one, one, one, None, one, 1 << 30, None) == -1
    d = _hip.BeamDesc(ctypes.sizeof(_hip.BeamDesc), 32, 125, 1024, 512, 40, 1, 0, 0, 0, 0, 16, 100, 1.0, 1.0)
    assert lib.nabu_speller_beam_ws_bytes(ctypes.byref(d)) > 32 * 16 * 125 * 1024 * 4      # tiled encoder output
    d.temperature = 0.0
    assert lib.nabu_speller_beam_ws_bytes(ctypes.byref(d)) == 0 and b'temperature' in lib.nabu_last_error()
    d.temperature, d.prob_fn = 1.0, 7
    assert lib.nabu_speller_beam_ws_bytes(ctypes.byref(d)) == 0 and b'probability_fn' in lib.nabu_last_error()
    a = _hip.AttnDesc(ctypes.sizeof(_hip.AttnDesc), 4, 10, 8, 8, 2, 1, 0, 0)    # windowed, right width 0
    assert lib.nabu_attn_fwd(ctypes.byref(a), 0, one, one, one, one, one, one, None, None, one, one, one, one,
                             None, None, 0, None) == -1
    assert b'right_window_width' in lib.nabu_last_error()
    a = _hip.AttnDesc(ctypes.sizeof(_hip.AttnDesc), 4, 10, 8, 8, 0, 0, 0, 3)    # unknown probability_fn
    assert lib.nabu_attn_fwd(ctypes.byref(a), 0, one, one, one, one, one, one, None, None, one, one, one, one,
                             None, None, 0, None) == -1
    assert b'probability_fn' in lib.nabu_last_error()
