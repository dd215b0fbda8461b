"""Pins the NumPy oracle (oracle/nabu_oracle.py) — the reference ships no tests or
golden vectors for this path, so the oracle is checked
against independent implementations available in the container:
PyTorch-CPU float64 autograd restatements (tests/torch_ref.py),
torch.nn.functional.ctc_loss, torch.nn.LSTM, finite differences and closed
forms."""
import os

import numpy as np
import pytest
import torch

from oracle import nabu_oracle as O
from tests import torch_ref as R

F64 = torch.float64


def _rng(seed):
    return np.random.default_rng(seed)


def _blstm_params(rng, D, H, scale=0.3):
    return dict(fw_kernel=rng.normal(0, scale, (D + H, 4 * H)),
                fw_bias=rng.normal(0, scale, 4 * H),
                bw_kernel=rng.normal(0, scale, (D + H, 4 * H)),
                bw_bias=rng.normal(0, scale, 4 * H))


def _t(p, grad=True):
    if isinstance(p, dict):
        return {k: _t(v, grad) for k, v in p.items()}
    if isinstance(p, list):
        return [_t(v, grad) for v in p]
    return torch.tensor(p, requires_grad=grad)


@pytest.mark.parametrize('reverse', [False, True])
def test_lstm_dir_matches_torch_autograd(reverse):
    rng = _rng(1)
    B, T, D, H = 4, 9, 5, 6
    x = rng.normal(size=(B, T, D))
    lens = np.array([9, 4, 1, 7])
    k = rng.normal(0, 0.4, (D + H, 4 * H))
    b = rng.normal(0, 0.4, 4 * H)
    out, cache = O.lstm_dir_fwd(x, lens, k, b, reverse)
    dout = rng.normal(size=out.shape)
    dx, dk, db = O.lstm_dir_bwd(dout, cache)
    xt, kt, bt = _t(x), _t(k), _t(b)
    o2 = R.lstm_dir(xt, lens, kt, bt, reverse)
    np.testing.assert_allclose(out, o2.detach().numpy(), atol=1e-12)
    (o2 * torch.tensor(dout)).sum().backward()
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(dk, kt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-11)
    # padded frames: zero output, zero input gradient
    for i, n in enumerate(lens):
        assert np.all(out[i, n:] == 0) and np.all(dx[i, n:] == 0)


def test_blstm_matches_torch_nn_lstm():
    """Independent implementation: torch.nn.LSTM(bidirectional) on packed
    sequences with gate blocks permuted (TF i,j,f,o -> torch i,f,g,o) and the
    +1 forget bias folded into the bias."""
    rng = _rng(2)
    B, T, D, H = 3, 7, 4, 5
    x = rng.normal(size=(B, T, D))
    lens = np.array([7, 5, 2])
    p = _blstm_params(rng, D, H)
    out, _ = O.blstm_fwd(x, lens, p)
    lstm = torch.nn.LSTM(D, H, batch_first=True, bidirectional=True).double()

    def conv(kern, bias, sfx):
        ki, kj, kf, ko = np.split(kern, 4, 1)
        bi, bj, bf, bo = np.split(bias, 4)
        w = np.concatenate([ki, kf, kj, ko], 1)          # torch order i,f,g,o
        bb = np.concatenate([bi, bf + 1.0, bj, bo])
        getattr(lstm, 'weight_ih_l0' + sfx).data = torch.tensor(w[:D].T.copy())
        getattr(lstm, 'weight_hh_l0' + sfx).data = torch.tensor(w[D:].T.copy())
        getattr(lstm, 'bias_ih_l0' + sfx).data = torch.tensor(bb)
        getattr(lstm, 'bias_hh_l0' + sfx).data = torch.zeros(4 * H, dtype=F64)
    conv(p['fw_kernel'], p['fw_bias'], '')
    conv(p['bw_kernel'], p['bw_bias'], '_reverse')
    pk = torch.nn.utils.rnn.pack_padded_sequence(torch.tensor(x), torch.tensor(lens),
                                                 batch_first=True, enforce_sorted=True)
    y, _ = lstm(pk)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(y, batch_first=True, total_length=T)
    np.testing.assert_allclose(out, y.detach().numpy(), atol=1e-12)


def test_lstm_zero_weights_closed_form():
    """kernel = 0, bias = 0: i=o=0.5, g=0, f=sigmoid(1) -> c stays 0, h = 0."""
    x = _rng(3).normal(size=(2, 5, 3))
    out, _ = O.lstm_dir_fwd(x, np.array([5, 3]), np.zeros((7, 16)), np.zeros(16), False)
    assert np.all(out == 0)
    # bias on j only: c_t = sum_k sigmoid(1)^k * 0.5*tanh(bj)
    bias = np.zeros(16); bias[4:8] = 0.7
    out, _ = O.lstm_dir_fwd(x, np.array([5, 3]), np.zeros((7, 16)), bias, False)
    f = 1 / (1 + np.exp(-1.0)); c = 0.0
    for t in range(5):
        c = c * f + 0.5 * np.tanh(0.7)
        np.testing.assert_allclose(out[0, t], np.tanh(c) * 0.5, atol=1e-14)


@pytest.mark.parametrize('T', [8, 7])
def test_pyramid_stack(T):
    rng = _rng(4)
    x = rng.normal(size=(3, T, 4))
    lens = np.array([T, 3, 1])
    o, l = O.pyramid_stack_fwd(x, lens, 2)
    o2, l2 = R.pyramid(torch.tensor(x), lens, 2)
    np.testing.assert_array_equal(o, o2.numpy())
    assert list(l) == l2
    d = rng.normal(size=o.shape)
    dx = O.pyramid_stack_bwd(d, T, 2)
    xt = torch.tensor(x, requires_grad=True)
    (R.pyramid(xt, lens, 2)[0] * torch.tensor(d)).sum().backward()
    np.testing.assert_array_equal(dx, xt.grad.numpy())


@pytest.mark.parametrize('T', [12, 11])
def test_listener_ctc_end_to_end_grads(T):
    rng = _rng(5)
    B, D, H, C = 3, 4, 5, 6
    x = rng.normal(size=(B, T, D))
    lens = np.array([T, 9, 8])
    layers = [_blstm_params(rng, D, H), _blstm_params(rng, 4 * H, H), _blstm_params(rng, 4 * H, H)]
    W = rng.normal(0, 0.5, (2 * H, C)); bo = rng.normal(0, 0.1, C)
    labels = [[0, 1], [2, 2], [4]]
    llen = np.array([2, 2, 1])
    enc, el, caches = O.listener_fwd(x, lens, layers)
    lg = O.linear_fwd(enc, W, bo)
    nll, dlg = O.ctc_loss(lg, el, labels, llen)
    dlg = dlg / B
    denc, dW, dbo = O.linear_bwd(dlg, enc, W)
    dx, grads = O.listener_bwd(denc, caches)

    xt, lt, Wt, bt = _t(x), _t(layers), _t(W), _t(bo)
    e2, el2 = R.listener(xt, lens, lt)
    assert list(el) == el2
    lg2 = e2 @ Wt + bt
    nll2 = R.ctc_mean(lg2, el2, labels, llen)
    np.testing.assert_allclose(nll, nll2.detach().numpy(), rtol=1e-10)
    nll2.mean().backward()
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dW, Wt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dbo, bt.grad.numpy(), atol=1e-10)
    for g, l in zip(grads, lt):
        for k in g:
            np.testing.assert_allclose(g[k], l[k].grad.numpy(), atol=1e-10)


def test_dblstm_matches_torch():
    rng = _rng(6)
    B, T, D, H = 2, 6, 3, 4
    x = rng.normal(size=(B, T, D)); lens = np.array([6, 4])
    layers = [_blstm_params(rng, D, H), _blstm_params(rng, 2 * H, H)]
    o, l, caches = O.dblstm_fwd(x, lens, layers)
    d = rng.normal(size=o.shape)
    dx, grads = O.dblstm_bwd(d, caches)
    xt, lt = _t(x), _t(layers)
    o2, _ = R.dblstm(xt, lens, lt)
    np.testing.assert_allclose(o, o2.detach().numpy(), atol=1e-12)
    (o2 * torch.tensor(d)).sum().backward()
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-11)
    for g, lp in zip(grads, lt):
        for k in g:
            np.testing.assert_allclose(g[k], lp[k].grad.numpy(), atol=1e-11)


def test_ctc_vs_torch_ctc_loss():
    rng = _rng(7)
    B, T, C = 5, 14, 7
    logits = rng.normal(0, 2, (B, T, C))
    tl = np.array([14, 10, 6, 14, 3])
    labels = [[0, 1, 1, 2], [5, 5, 5], [3], [], [0, 0]]
    ll = np.array([4, 3, 1, 0, 2])
    nll, g = O.ctc_loss(logits, tl, labels, ll)
    lt = torch.tensor(logits, requires_grad=True)
    # torch cannot take an empty target inside a concatenated batch reliably: do per-utterance
    tot = 0
    for b in range(B):
        lp = torch.log_softmax(lt[b:b + 1, :tl[b]], 2).transpose(0, 1)
        v = torch.nn.functional.ctc_loss(lp, torch.tensor([labels[b]], dtype=torch.long).reshape(1, -1),
                                         torch.tensor([tl[b]]), torch.tensor([ll[b]]),
                                         blank=C - 1, reduction='none')
        np.testing.assert_allclose(nll[b], v.item(), rtol=1e-10)
        tot = tot + v.sum()
    tot.backward()
    np.testing.assert_allclose(g, lt.grad.numpy(), atol=1e-10)
    for b in range(B):
        assert np.all(g[b, tl[b]:] == 0)


def test_ctc_closed_forms():
    C = 5
    rng = _rng(8)
    # T=1, L=1: -log softmax(label)
    lg = rng.normal(size=(1, 1, C))
    nll, _ = O.ctc_loss(lg, [1], [[2]], [1])
    p = np.exp(lg[0, 0]) / np.exp(lg[0, 0]).sum()
    np.testing.assert_allclose(nll[0], -np.log(p[2]), rtol=1e-12)
    # all-blank target: -sum_t log p_blank
    lg = rng.normal(size=(1, 6, C))
    nll, _ = O.ctc_loss(lg, [6], [[]], [0])
    ls = lg[0] - np.log(np.exp(lg[0]).sum(1, keepdims=True))
    np.testing.assert_allclose(nll[0], -ls[:, C - 1].sum(), rtol=1e-12)
    # uniform logits: p = (#alignments) * C^-T ; label "a" over T frames with
    # blank: alignments = number of (start,end) runs of a = T(T+1)/2
    T = 5
    nll, _ = O.ctc_loss(np.zeros((1, T, C)), [T], [[0]], [1])
    np.testing.assert_allclose(nll[0], -np.log(T * (T + 1) / 2 * C ** -T), rtol=1e-12)
    # infeasible: repeated label needs a blank in between
    with pytest.raises(ValueError):
        O.ctc_loss(np.zeros((1, 2, C)), [2], [[1, 1]], [2])


def test_ctc_finite_difference():
    rng = _rng(9)
    lg = rng.normal(size=(1, 6, 4))
    lab, ll = [[0, 2, 2]], [3]
    _, g = O.ctc_loss(lg, [6], lab, ll)
    eps = 1e-6
    for idx in [(0, 0, 0), (0, 3, 2), (0, 5, 3)]:
        a = lg.copy(); a[idx] += eps
        b = lg.copy(); b[idx] -= eps
        fd = (O.ctc_loss(a, [6], lab, ll)[0][0] - O.ctc_loss(b, [6], lab, ll)[0][0]) / (2 * eps)
        np.testing.assert_allclose(g[idx], fd, atol=1e-7)


def test_average_cross_entropy_vs_torch():
    rng = _rng(10)
    B, L, C = 3, 6, 5
    lg = rng.normal(size=(B, L, C))
    tg = rng.integers(0, C, (B, L + 2))
    tl = np.array([6, 4, 2])
    loss, g = O.average_cross_entropy(lg, tg, tl, tl)
    lt = torch.tensor(lg, requires_grad=True)
    l2 = R.avg_xent(lt, tg, tl, tl)
    np.testing.assert_allclose(loss, l2.item(), rtol=1e-12)
    l2.backward()
    np.testing.assert_allclose(g, lt.grad.numpy(), atol=1e-12)


def _speller_params(rng, E, U, C, nl, attention, K=5, F=3):
    p = dict(memory_kernel=rng.normal(0, 0.3, (E, U)), query_kernel=rng.normal(0, 0.3, (U, U)),
             attention_v=rng.normal(0, 0.5, U), out_kernel=rng.normal(0, 0.3, (U + E, C)),
             out_bias=rng.normal(0, 0.1, C), lstm=[])
    for n in range(nl):
        nin = (C + E) if n == 0 else U
        p['lstm'].append(dict(kernel=rng.normal(0, 0.3, (nin + U, 4 * U)),
                              bias=rng.normal(0, 0.1, 4 * U)))
    if attention == 'location_aware':
        p['conv_kernel'] = rng.normal(0, 0.5, (K, F))
        p['conv_proj'] = rng.normal(0, 0.5, (F, U))
    return p


@pytest.mark.parametrize('attention,prob_fn,nl,K', [
    ('vanilla', 'softmax', 1, 0), ('vanilla', 'softmax', 2, 0),
    ('location_aware', 'softmax', 1, 5), ('location_aware', 'softmax', 2, 4),
    ('vanilla', 'sigmoid', 1, 0), ('location_aware', 'normalized_sigmoid', 1, 3),
    ('windowed', 'softmax', 1, 0), ('windowed', 'softmax', 2, 1), ('windowed', 'normalized_sigmoid', 1, 1),
    ('windowed', 'sigmoid', 1, 0)])
def test_speller_matches_torch_autograd(attention, prob_fn, nl, K):
    rng = _rng(11)
    B, Te, E, U, C = 3, 7, 6, 5, 6
    enc = rng.normal(size=(B, Te, E))
    enc_len = np.array([7, 5, 3])
    tl = np.array([5, 3, 4])
    targets = rng.integers(0, C - 1, (B, 5))
    for b in range(B):
        targets[b, tl[b] - 1] = C - 1       # eos
    p = _speller_params(rng, E, U, C, nl, attention, K=K)
    window = (K, 2) if attention == 'windowed' else None      # (left_window_width, right_window_width)
    lg, ll, cache = O.speller_fwd(enc, enc_len, targets, tl, p, attention, prob_fn, window=window)
    if window:
        # the window moves and really masks something
        masks = np.array([st['wmask'] for st in cache['steps']])
        assert masks[0, :, :2].all() and not masks[0, :, 2:].any()      # one-hot start: frames [0, right)
        assert (masks[1:] != masks[:1]).any()
    loss, dlg = O.average_cross_entropy(lg, targets, ll, tl)
    denc, g = O.speller_bwd(dlg, cache)
    et, pt = _t(enc), _t(p)
    lg2 = R.speller(et, enc_len, targets, tl, pt, attention, prob_fn, window)
    np.testing.assert_allclose(lg, lg2.detach().numpy(), atol=1e-11)
    l2 = R.avg_xent(lg2, targets, tl, tl)
    np.testing.assert_allclose(loss, l2.item(), rtol=1e-11)
    l2.backward()
    np.testing.assert_allclose(denc, et.grad.numpy(), atol=1e-11)
    for k in g:
        if k == 'lstm':
            for a, b_ in zip(g['lstm'], pt['lstm']):
                np.testing.assert_allclose(a['kernel'], b_['kernel'].grad.numpy(), atol=1e-11)
                np.testing.assert_allclose(a['bias'], b_['bias'].grad.numpy(), atol=1e-11)
        else:
            np.testing.assert_allclose(g[k], pt[k].grad.numpy(), atol=1e-11, err_msg=k)


def test_attention_equal_keys_gives_uniform_alignment():
    rng = _rng(12)
    B, Te, E, U, C = 2, 6, 4, 3, 5
    row = rng.normal(size=(1, 1, E))
    enc = np.repeat(np.repeat(row, Te, 1), B, 0)
    p = _speller_params(rng, E, U, C, 1, 'vanilla')
    _, _, cache = O.speller_fwd(enc, [6, 4], np.zeros((B, 2), int), [2, 2], p)
    np.testing.assert_allclose(cache['steps'][0]['al'][0], np.full(6, 1 / 6), atol=1e-14)
    np.testing.assert_allclose(cache['steps'][1]['al'][1], [0.25] * 4 + [0, 0], atol=1e-14)


def test_adam_and_lr_schedule():
    rng = _rng(13)
    th = rng.normal(size=50); g = rng.normal(0, 2, 50)
    m = np.zeros(50); v = np.zeros(50)
    t1, m1, v1 = O.clip_adam_update(th, g, m, v, 1, 1e-3)
    gc = np.clip(g, -1, 1)
    # first step closed form: lr_t*m/(sqrt(v)+eps) with m=(1-b1)g, v=(1-b2)g^2
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(t1, th - lr_t * 0.1 * gc / (np.sqrt(0.001) * np.abs(gc) + 1e-8), rtol=1e-12)
    # against torch.optim.Adam on a case where eps is negligible
    tt = torch.tensor(th, requires_grad=True)
    opt = torch.optim.Adam([tt], lr=1e-3, eps=1e-30)
    mm, vv, cur = m, v, th
    for step in range(1, 4):
        opt.zero_grad(); tt.grad = torch.tensor(gc * step); opt.step()
        cur, mm, vv = O.clip_adam_update(cur, gc * step, mm, vv, step, 1e-3, clip=1e9, eps=1e-30)
    np.testing.assert_allclose(cur, tt.detach().numpy(), rtol=1e-9)
    assert O.learning_rate(1e-3, 0.1, 0, 100) == 1e-3
    np.testing.assert_allclose(O.learning_rate(1e-3, 0.1, 50, 100, 0.5), 1e-3 * 0.1 ** 0.5 * 0.5)


def test_relu_layer_norm_vs_torch_autograd():
    """DNNDecoder hidden-layer pieces: contrib layer_norm normalises over (T,F) per batch row"""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(3, 5, 4))
    gamma, beta = rng.normal(size=4), rng.normal(size=4)
    dy = rng.normal(size=x.shape)
    y, cache = O.layer_norm_fwd(x, gamma, beta)
    dx, dg, db = O.layer_norm_bwd(dy, cache)
    tx = torch.tensor(x, dtype=F64, requires_grad=True)
    tg = torch.tensor(gamma, dtype=F64, requires_grad=True)
    tb = torch.tensor(beta, dtype=F64, requires_grad=True)
    mu = tx.reshape(3, -1).mean(1).reshape(3, 1, 1)
    var = ((tx - mu) ** 2).reshape(3, -1).mean(1).reshape(3, 1, 1)
    ty = (tx - mu) / torch.sqrt(var + 1e-12) * tg + tb
    np.testing.assert_allclose(ty.detach().numpy(), y, atol=1e-12)
    (ty * torch.tensor(dy)).sum().backward()
    np.testing.assert_allclose(tx.grad.numpy(), dx, atol=1e-11)
    np.testing.assert_allclose(tg.grad.numpy(), dg, atol=1e-11)
    np.testing.assert_allclose(tb.grad.numpy(), db, atol=1e-11)
    r = O.relu_fwd(x)
    assert np.array_equal(O.relu_bwd(dy, r), np.where(x > 0, dy, 0))


def test_cpu_baseline_restatement_is_the_same_graph():
    """oracle/cpu_baseline.py (what bench.py times on the host: the reference graph at TF op granularity in
    PyTorch-CPU) computes what the oracle computes — a ragged BLSTM layer in both its per-frame form and its
    torch.nn.LSTM form, values and input gradient, in float64"""
    from oracle import cpu_baseline as CB
    rng = np.random.default_rng(11)
    B, T, D, H = 3, 9, 5, 4
    lens = np.array([9, 6, 2], np.int32)
    x = rng.normal(size=(B, T, D))
    x *= (np.arange(T)[None, :, None] < lens[:, None, None])
    p = {'%s_%s' % (d, k): rng.normal(size=s) * 0.5 for d in ('fw', 'bw')
         for k, s in (('kernel', (D + H, 4 * H)), ('bias', (4 * H,)))}
    y, cache = O.blstm_fwd(x, lens, p)
    dy = rng.normal(size=y.shape)
    dx, _ = O.blstm_bwd(dy, cache)
    tl = torch.tensor(lens).long()
    for fn in (CB.dynamic_rnn, CB.fused_lstm_dir):
        xt = torch.tensor(x, requires_grad=True)
        yt = torch.cat([fn(xt, tl, torch.tensor(p['fw_kernel']), torch.tensor(p['fw_bias']), False),
                        fn(xt, tl, torch.tensor(p['bw_kernel']), torch.tensor(p['bw_bias']), True)], 2)
        np.testing.assert_allclose(yt.detach().numpy(), y, atol=1e-12)
        (yt * torch.tensor(dy)).sum().backward()
        np.testing.assert_allclose(xt.grad.numpy(), dx, atol=1e-12)
    # and one whole training step of the timed configuration runs and decreases nothing silently: same loss
    # from the per-frame and the fused form on identical weights
    layers, out = CB.make_params('cfg1')
    batch = CB.make_batch('cfg1')
    small = (batch[0][:2, :30], torch.tensor([30, 22]), batch[2][:2, :5], torch.tensor([5, 4]))
    l1 = CB.train_step('cfg1', [{k: v.detach().clone().requires_grad_() for k, v in l.items()} for l in layers],
                       {k: v.detach().clone().requires_grad_() for k, v in out.items()}, small, {'t': 0, 'm': {}, 'v': {}})
    l2 = CB.train_step('cfg1', [{k: v.detach().clone().requires_grad_() for k, v in l.items()} for l in layers],
                       {k: v.detach().clone().requires_grad_() for k, v in out.items()}, small, {'t': 0, 'm': {}, 'v': {}},
                       fused=True)
    assert abs(l1 - l2) < 1e-4 * abs(l1)


def test_ctc_matches_tensorflows_own_known_answer_vectors():
    """[TF-1.8 recalled] The two sequences of TensorFlow's own CTC unit test (tensorflow/python/kernel_tests/
    ctc_loss_op_test.py, CTCLossTest.testBasic: 6 classes, 5 frames, targets [0,1,2,1,0] and [0,1,1,0], inputs =
    log of the probability matrices, expected losses 3.34211 / 5.42262 and expected d loss / d logits), restated
    from memory as tests/golden/tf_ctc_basic.npz.  They are not held by the reference repository (it has no CTC
    test), so parity stays "unpinned" under the rubric; but 62 recalled numbers the oracle reproduces to their
    printed precision cannot agree by chance: they pin the blank index (last class), the handling of the repeated
    label in the second target and the sign / normalisation of the gradient of tf.nn.ctc_loss
    (reference call site: nabu/neuralnetworks/trainers/loss_functions.py:206-210)."""
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_ctc_basic.npz'))
    nll, grad = O.ctc_loss(np.log(fx['prob']), fx['logit_len'], fx['labels'], fx['label_len'])
    assert np.abs(nll - fx['loss']).max() < 1e-5
    assert np.abs(grad - fx['grad']).max() < 2e-6
