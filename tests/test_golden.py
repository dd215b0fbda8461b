"""CPU: the committed golden fixtures (tests/golden/*.npz) are (1) what the float64 oracle
produces today — so the checker cannot drift silently — and (2) confirmed by an independent
implementation (torch float64 autograd, tests/torch_ref.py).  The reference itself (TF-1.8)
cannot run here: parity is unpinned against TF (oracle header)."""
import os

import numpy as np
import pytest
import torch

from oracle import nabu_oracle as O
from tests import torch_ref as R
from tests.golden import make_golden as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
F64 = torch.float64


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def unpack(fx, prefix):
    return {k[len(prefix):].replace('|', '/'): v for k, v in fx.items() if k.startswith(prefix)}


def batch_of(fx, s):
    return dict(inputs={'features': fx['x%d' % s]}, input_seq_length={'features': fx['xl%d' % s]},
                targets={'text': fx['y%d' % s]}, target_seq_length={'text': fx['yl%d' % s]})


def test_components_ctc_fixture_matches_oracle_and_torch():
    fx = load('components')
    nll, dlg = O.ctc_loss(fx['ctc_logits'], fx['ctc_len'], fx['ctc_labels'], fx['ctc_label_len'])
    np.testing.assert_allclose(nll, fx['ctc_nll'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(dlg, fx['ctc_grad'], rtol=0, atol=1e-12)
    lg = torch.tensor(fx['ctc_logits'], dtype=F64, requires_grad=True)
    tn = R.ctc_mean(lg, fx['ctc_len'], fx['ctc_labels'], fx['ctc_label_len'])
    np.testing.assert_allclose(tn.detach().numpy(), fx['ctc_nll'], rtol=1e-10)
    tn.sum().backward()
    np.testing.assert_allclose(lg.grad.numpy(), fx['ctc_grad'], atol=1e-10)
    # empty label sequence: -sum log p(blank) over the valid frames (closed form)
    lp = fx['ctc_logits'][3, :3] - np.log(np.exp(fx['ctc_logits'][3, :3]).sum(1, keepdims=True))
    assert abs(fx['ctc_nll'][3] + lp[:, -1].sum()) < 1e-12


def test_components_blstm_fixture_matches_oracle_and_torch():
    fx = load('components')
    p = {k[len('lstm_p_'):]: v for k, v in fx.items() if k.startswith('lstm_p_')}
    y, cache = O.blstm_fwd(fx['lstm_x'], fx['lstm_len'], p)
    np.testing.assert_allclose(y, fx['lstm_out'], atol=1e-13)
    dx, g = O.blstm_bwd(fx['lstm_dout'], cache)
    np.testing.assert_allclose(dx, fx['lstm_dx'], atol=1e-13)
    x = torch.tensor(fx['lstm_x'], dtype=F64, requires_grad=True)
    tp = {k: torch.tensor(v, dtype=F64, requires_grad=True) for k, v in p.items()}
    ty = R.blstm(x, fx['lstm_len'], tp)
    np.testing.assert_allclose(ty.detach().numpy(), fx['lstm_out'], atol=1e-12)
    (ty * torch.tensor(fx['lstm_dout'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), fx['lstm_dx'], atol=1e-12)
    for k in p:
        np.testing.assert_allclose(tp[k].grad.numpy(), fx['lstm_g_' + k], atol=1e-12)
    assert np.all(fx['lstm_out'][1, 4:] == 0) and np.all(fx['lstm_out'][2, 1:] == 0)


def test_components_adam_fixture():
    fx = load('components')
    th, m, v = fx['adam_theta0'].copy(), np.zeros(50), np.zeros(50)
    for t in range(3):
        g = np.clip(fx['adam_grads'][t], -1, 1)                   # trainer.py:560-563
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** (t + 1)) / (1 - 0.9 ** (t + 1))
        th = th - lr_t * m / (np.sqrt(v) + 1e-8)                  # TF: eps outside the bias correction
    np.testing.assert_allclose(th, fx['adam_theta'], atol=1e-14)
    np.testing.assert_allclose(m, fx['adam_m'], atol=1e-15)
    np.testing.assert_allclose(v, fx['adam_v'], atol=1e-15)


@pytest.mark.parametrize('name,enc,nl', [('cfg1_small', 'DBLSTM', 2), ('cfg2_small', 'Listener', 3)])
def test_small_ctc_fixture_matches_oracle_and_torch(name, enc, nl):
    fx = load(name)
    w = unpack(fx, 'w:')
    loss, g = G.step_ctc(w, batch_of(fx, 0), enc, nl)
    assert abs(loss - fx['losses'][0]) < 1e-9
    want = unpack(fx, 'g:')
    for k in want:
        np.testing.assert_allclose(g[k], want[k].reshape(g[k].shape), rtol=2e-5, atol=1e-7)   # stored as f32
    # independent: torch autograd on the same weights and batch
    layers = G.enc_layers(w, enc, nl)
    tl = [{k: torch.tensor(v, dtype=F64, requires_grad=True) for k, v in l.items()} for l in layers]
    x = torch.tensor(fx['x0'], dtype=F64)
    e, el = (R.listener if enc == 'Listener' else R.dblstm)(x, fx['xl0'], tl)
    W = torch.tensor(w['DNNDecoder/text/outlayer/weights'], dtype=F64, requires_grad=True)
    b = torch.tensor(w['DNNDecoder/text/outlayer/biases'], dtype=F64, requires_grad=True)
    tloss = R.ctc_mean(e @ W + b, np.asarray(el, np.int64), fx['y0'], fx['yl0']).mean()
    assert abs(float(tloss.detach()) - fx['losses'][0]) / fx['losses'][0] < 1e-10
    tloss.backward()
    np.testing.assert_allclose(W.grad.numpy(), want['DNNDecoder/text/outlayer/weights'], rtol=2e-4, atol=1e-6)
    k0 = [k for k in want if k.endswith('layer0/' + ('BLSTM/' if enc == 'Listener' else '') + G.CELL % ('bw', 'kernel'))][0]
    np.testing.assert_allclose(tl[0]['bw_kernel'].grad.numpy(), want[k0], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize('name,attention', [('cfg3_small', 'vanilla'), ('cfg5_small', 'location_aware')])
def test_small_las_fixture_matches_oracle_and_torch(name, attention):
    fx = load(name)
    w = unpack(fx, 'w:')
    loss, g = G.step_las(w, batch_of(fx, 0), 3, 1, attention)
    assert abs(loss - fx['losses'][0]) < 1e-10
    want = unpack(fx, 'g:')
    for k in want:
        np.testing.assert_allclose(g[k].reshape(want[k].shape), want[k], rtol=2e-5, atol=1e-8)
    layers = G.enc_layers(w, 'Listener', 3)
    tl = [{k: torch.tensor(v, dtype=F64) for k, v in l.items()} for l in layers]
    e, el = R.listener(torch.tensor(fx['x0'], dtype=F64), fx['xl0'], tl)
    p = G.speller_view(w, 1, attention)

    def tt(x):
        if isinstance(x, dict):
            return {k: tt(v) for k, v in x.items()}
        if isinstance(x, list):
            return [tt(v) for v in x]
        return torch.tensor(x, dtype=F64, requires_grad=True)
    tp = tt(p)
    lg = R.speller(e, el, fx['y0'], fx['yl0'], tp, attention, 'softmax')
    tloss = R.avg_xent(lg, fx['y0'], fx['yl0'], fx['yl0'])
    assert abs(float(tloss.detach()) - fx['losses'][0]) / fx['losses'][0] < 1e-10
    tloss.backward()
    np.testing.assert_allclose(tp['attention_v'].grad.numpy(),
                               want[[k for k in want if k.endswith('attention_v')][0]], rtol=2e-4, atol=1e-8)


def test_cfg1_exact_fixture_matches_oracle():
    """BASELINE.json configs[0] at full size: 1 step re-computed (the fixture holds 3)"""
    fx = load('cfg1_exact')
    names, data = G.cfg1_exact_setup()
    w = G.draw_weights(names)
    loss, g = G.step_ctc(w, data.batch(0), 'DBLSTM', 2)
    assert abs(loss - fx['losses'][0]) < 1e-8
    for k, gr in g.items():
        flat = np.asarray(gr).ravel()
        key = k.replace('/', '|')
        assert abs(np.sqrt((flat ** 2).sum()) - fx['gnorm:' + key]) < 1e-9 * max(1.0, fx['gnorm:' + key])
        np.testing.assert_allclose(flat[G.sample_index(k, flat.size)], fx['gsample:' + key], atol=1e-12)


def test_full_size_fixtures_carry_their_independent_cross_check():
    """cfg2_exact / cfg3_exact / cfg5_exact (BASELINE configs at FULL size; minutes of float64 per
    config, so they are not recomputed in every CPU run): the outcome of the one-time independent
    recomputation by torch float64 autograd (tests/golden/xcheck_exact.py) is committed next to
    them — loss to 1e-9, every gradient norm and the sampled gradient entries to 1e-7 relative —
    and the fixture's loss is the number that run compared with.  NABU_SLOW_TESTS=1 recomputes
    the headline config with the oracle."""
    import json
    with open(os.path.join(GOLD, 'exact_xcheck.json')) as fid:
        xc = json.load(fid)
    for name, nvars in (('cfg2_exact', 18), ('cfg3_exact', 23), ('cfg5_exact', 25)):
        fx = load(name)
        r = xc[name]
        assert r['variables'] == nvars == len([k for k in fx if k.startswith('gnorm:')])
        assert r['loss_rel'] < 1e-9 and r['gnorm_rel_max'] < 1e-7 and r['gsample_rel_max'] < 1e-7, (name, r)
        assert r['loss_oracle'] == float(fx['losses'][0])
    if os.environ.get('NABU_SLOW_TESTS') == '1':
        names, data, step_fn, _, _, _ = G.exact_setup('cfg2_exact')
        loss, g = step_fn(G.draw_weights(names), data.batch(0))
        fx = load('cfg2_exact')
        assert abs(loss - fx['losses'][0]) < 1e-8
        for k, gr in g.items():
            flat = np.asarray(gr).ravel()
            np.testing.assert_allclose(flat[G.sample_index(k, flat.size)], fx['gsample:' + k.replace('/', '|')],
                                       atol=1e-12)


def test_decode_fixture_matches_the_decode_oracle():
    """tests/golden/decode.npz (inference) is what oracle/decode_oracle.py gives"""
    from oracle import decode_oracle as DO
    fx = load('decode')
    for merge in (1, 0):
        hyps = DO.ctc_decode_batch(fx['ctc_logits'], fx['ctc_lens'], 100, bool(merge))
        for b, h in enumerate(hyps):
            n = fx['ctc_len_merge%d' % merge][b]
            assert list(fx['ctc_ids_merge%d' % merge][b, :n]) == h
    hyps = DO.ctc_decode_batch(fx['ctc_logits'], fx['ctc_lens'], 100, True)
    for b, h in enumerate(hyps):
        assert DO.edit_distance(h, list(fx['ed_ref'][b, :fx['ed_ref_len'][b]])) == fx['ed_dist'][b]
    for attention in ('vanilla', 'location_aware'):
        pre = 'bs_%s_' % attention
        w = unpack(fx, pre + 'w_')
        res = DO.speller_beam_search(fx[pre + 'enc'].astype(np.float64), fx[pre + 'enc_len'],
                                     G.speller_view(w, 1, attention), 6, 12, 1.0, 1.0, attention)
        np.testing.assert_array_equal(res['sequences'], fx[pre + 'seq'])
        np.testing.assert_array_equal(res['lengths'], fx[pre + 'len'])
        np.testing.assert_allclose(res['scores'], fx[pre + 'scores'], rtol=1e-6)
