"""Generates the golden fixtures of tests/golden/ from the float64 oracle.

    python tests/golden/make_golden.py            # rewrites every *.npz next to this file

PROVENANCE.  The reference (TF-1.8 / Python 2) can be neither run nor imported in the build
container; these vectors therefore come from oracle/nabu_oracle.py — the CPU
restatement of the reference's algorithm — and NOT from the reference itself: parity is
UNPINNED against TF.  What the fixtures buy: (1) they freeze the oracle (tests/test_golden.py
re-computes them on every CPU run, so an accidental change of the checker is caught),
(2) they are cross-checked by an independent implementation (torch float64 autograd,
tests/torch_ref.py) in the same CPU test, (3) the GPU parity tests can compare the HIP path
with committed numbers without executing the oracle.

Every fixture holds inputs (features, lengths, labels), the weights under the reference's
TF variable names, and the expected loss, gradients and loss
trajectory under clip + Adam.  cfg1 at its exact BASELINE size regenerates inputs and weights
from seeds (documented below) and stores the expected losses plus gradient norms and sampled
gradient entries, to keep the file small.

Generators: batches = nabu_amd.processing.synthetic.SyntheticData (numpy PCG64,
a pure function of (seed, step)); weights = oracle.glorot_uniform(default_rng([99, i])) for the
i-th variable in the order listed by `*_names()` below (zeros for Dense / LSTMCell biases)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import nabu_oracle as O                                  # noqa: E402
from nabu_amd.processing.synthetic import SyntheticData              # noqa: E402

CELL = 'bidirectional_rnn/%s/layer_norm_basic_lstm_cell/%s'
SP = 'Speller/decoder/'
LR = 1e-3


# ---------------------------------------------------------------- weights by TF name
def encoder_names(enc, num_layers, D, H, pyramid=2):
    """[(name, shape)] in variable-creation order"""
    out = []
    n = num_layers + 1 if enc == 'Listener' else num_layers
    din = D
    for l in range(n):
        pyr = enc == 'Listener' and l < num_layers
        pre = '%s/features/layer%d/%s' % (enc, l, 'BLSTM/' if pyr else '')
        for d in ('fw', 'bw'):
            out.append((pre + CELL % (d, 'kernel'), (din + H, 4 * H)))
            out.append((pre + CELL % (d, 'bias'), (4 * H,)))
        din = 2 * H * (pyramid if pyr else 1)
    return out, din


def ctc_decoder_names(E, C):
    return [('DNNDecoder/text/outlayer/weights', (E, C)), ('DNNDecoder/text/outlayer/biases', (C,))]


def speller_names(E, U, C, nl, attention, K=0, F=0):
    sc = 'bahdanau_attention' if attention == 'vanilla' else 'location_aware_attention'
    out = [(SP + 'memory_layer/kernel', (E, U)), (SP + sc + '/query_layer/kernel', (U, U)),
           (SP + sc + '/attention_v', (U,))]
    if attention == 'location_aware':
        out += [(SP + sc + '/conv1d/kernel', (K, 1, F)), (SP + sc + '/process_conv_features/kernel', (F, U))]
    for n in range(nl):
        q = SP + 'attention_wrapper/multi_rnn_cell/cell_%d/lstm_cell/' % n
        out += [(q + 'kernel', ((C + E if n == 0 else U) + U, 4 * U)), (q + 'bias', (4 * U,))]
    out += [(SP + 'dense/kernel', (U + E, C)), (SP + 'dense/bias', (C,))]
    return out


ZERO_INIT = ('outlayer/biases', 'lstm_cell/bias', 'dense/bias')


def draw_weights(names):
    w = {}
    for i, (name, shape) in enumerate(names):
        if name.endswith(ZERO_INIT):
            w[name] = np.zeros(shape, np.float32)
        else:
            w[name] = O.glorot_uniform(np.random.default_rng([99, i]), shape)
    return w


# ---------------------------------------------------------------- oracle views of the weights
def enc_layers(w, enc, num_layers):
    n = num_layers + 1 if enc == 'Listener' else num_layers
    layers = []
    for l in range(n):
        pyr = enc == 'Listener' and l < num_layers
        pre = '%s/features/layer%d/%s' % (enc, l, 'BLSTM/' if pyr else '')
        layers.append({'%s_%s' % (d, k): w[pre + CELL % (d, k)].astype(np.float64)
                       for d in ('fw', 'bw') for k in ('kernel', 'bias')})
    return layers


def enc_grads_by_name(grads, enc, num_layers):
    out = {}
    for l, g in enumerate(grads):
        pyr = enc == 'Listener' and l < num_layers
        pre = '%s/features/layer%d/%s' % (enc, l, 'BLSTM/' if pyr else '')
        for d in ('fw', 'bw'):
            for k in ('kernel', 'bias'):
                out[pre + CELL % (d, k)] = g['%s_%s' % (d, k)]
    return out


def speller_view(w, nl, attention):
    sc = 'bahdanau_attention' if attention == 'vanilla' else 'location_aware_attention'
    f = lambda a: a.astype(np.float64)
    p = dict(memory_kernel=f(w[SP + 'memory_layer/kernel']), query_kernel=f(w[SP + sc + '/query_layer/kernel']),
             attention_v=f(w[SP + sc + '/attention_v']), out_kernel=f(w[SP + 'dense/kernel']),
             out_bias=f(w[SP + 'dense/bias']), lstm=[])
    for n in range(nl):
        q = SP + 'attention_wrapper/multi_rnn_cell/cell_%d/lstm_cell/' % n
        p['lstm'].append(dict(kernel=f(w[q + 'kernel']), bias=f(w[q + 'bias'])))
    if attention == 'location_aware':
        ck = f(w[SP + sc + '/conv1d/kernel'])
        p['conv_kernel'] = ck.reshape(ck.shape[0], ck.shape[2])
        p['conv_proj'] = f(w[SP + sc + '/process_conv_features/kernel'])
    return p


def speller_grads_by_name(g, w, nl, attention):
    sc = 'bahdanau_attention' if attention == 'vanilla' else 'location_aware_attention'
    out = {SP + 'memory_layer/kernel': g['memory_kernel'], SP + sc + '/query_layer/kernel': g['query_kernel'],
           SP + sc + '/attention_v': g['attention_v'], SP + 'dense/kernel': g['out_kernel'],
           SP + 'dense/bias': g['out_bias']}
    if attention == 'location_aware':
        out[SP + sc + '/conv1d/kernel'] = g['conv_kernel'].reshape(w[SP + sc + '/conv1d/kernel'].shape)
        out[SP + sc + '/process_conv_features/kernel'] = g['conv_proj']
    for n in range(nl):
        q = SP + 'attention_wrapper/multi_rnn_cell/cell_%d/lstm_cell/' % n
        out[q + 'kernel'] = g['lstm'][n]['kernel']
        out[q + 'bias'] = g['lstm'][n]['bias']
    return out


# ---------------------------------------------------------------- one training step on the oracle
def step_ctc(w, batch, enc, num_layers):
    layers = enc_layers(w, enc, num_layers)
    x = batch['inputs']['features'].astype(np.float64)
    lens = batch['input_seq_length']['features']
    fwd, bwd = (O.listener_fwd, O.listener_bwd) if enc == 'Listener' else (O.dblstm_fwd, O.dblstm_bwd)
    e, el, caches = fwd(x, lens, layers)
    W = w['DNNDecoder/text/outlayer/weights'].astype(np.float64)
    b = w['DNNDecoder/text/outlayer/biases'].astype(np.float64)
    lg = O.linear_fwd(e, W, b)
    nll, dlg = O.ctc_loss(lg, el, batch['targets']['text'], batch['target_seq_length']['text'])
    de, dW, db = O.linear_bwd(dlg / x.shape[0], e, W)
    _, grads = bwd(de, caches)
    g = enc_grads_by_name(grads, enc, num_layers)
    g['DNNDecoder/text/outlayer/weights'] = dW
    g['DNNDecoder/text/outlayer/biases'] = db
    return float(nll.mean()), g


def step_las(w, batch, num_layers, nl, attention):
    layers = enc_layers(w, 'Listener', num_layers)
    p = speller_view(w, nl, attention)
    enc, el, caches = O.listener_fwd(batch['inputs']['features'].astype(np.float64),
                                     batch['input_seq_length']['features'], layers)
    tg, tl = batch['targets']['text'], batch['target_seq_length']['text']
    lg, ll, cache = O.speller_fwd(enc, el, tg, tl, p, attention)
    loss, dlg = O.average_cross_entropy(lg, tg, ll, tl)
    denc, sg = O.speller_bwd(dlg, cache)
    _, gl = O.listener_bwd(denc, caches)
    g = enc_grads_by_name(gl, 'Listener', num_layers)
    g.update(speller_grads_by_name(sg, w, nl, attention))
    return float(loss), g


def trajectory(w, data, step_fn, steps):
    """clip + Adam (trainer.py:512-580) on successive batches; returns losses, step-0 grads"""
    w = {k: v.astype(np.float64) for k, v in w.items()}
    m = {k: np.zeros_like(v) for k, v in w.items()}
    v2 = {k: np.zeros_like(v) for k, v in w.items()}
    losses, g0 = [], None
    for s in range(steps):
        loss, g = step_fn(w, data.batch(s))
        losses.append(loss)
        if s == 0:
            g0 = g
        for k in w:
            w[k], m[k], v2[k] = O.clip_adam_update(w[k], g[k].reshape(w[k].shape), m[k], v2[k], s + 1, LR)
    return np.array(losses), g0, w


def pack(prefix, d, dtype=np.float32):
    return {prefix + k.replace('/', '|'): np.asarray(v, dtype) for k, v in d.items()}


def batch_arrays(data, steps):
    out = {}
    for s in range(steps):
        b = data.batch(s)
        out['x%d' % s] = b['inputs']['features']
        out['xl%d' % s] = b['input_seq_length']['features']
        out['y%d' % s] = b['targets']['text']
        out['yl%d' % s] = b['target_seq_length']['text']
    return out


# ---------------------------------------------------------------- fixtures
def make_small_ctc(name, enc, num_layers, H, B, T, red, seed, steps=4):
    D, C = 40, 40
    names, E = encoder_names(enc, num_layers, D, H)
    names += ctc_decoder_names(E if enc == 'Listener' else 2 * H, C)
    w = draw_weights(names)
    data = SyntheticData(B, T, D, min_frames=int(0.6 * T), min_labels=2, max_labels=5, time_reduction=red, seed=seed)
    losses, g0, wT = trajectory(w, data, lambda ww, b: step_ctc(ww, b, enc, num_layers), steps)
    out = dict(meta=np.array([B, T, D, H, num_layers, C, steps, seed]), losses=losses)
    out.update(batch_arrays(data, steps))
    out.update(pack('w:', w))
    out.update(pack('g:', g0))
    out.update(pack('wT:', wT))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    return losses


def make_small_las(name, attention, H, U, B, T, seed, K=0, F=0, steps=3):
    D, C, num_layers, nl = 40, 40, 3, 1
    names, E = encoder_names('Listener', num_layers, D, H)
    names += speller_names(E, U, C, nl, attention, K, F)
    w = draw_weights(names)
    data = SyntheticData(B, T, D, min_frames=int(0.6 * T), min_labels=2, max_labels=6, eos=True, time_reduction=8,
                         seed=seed)
    losses, g0, wT = trajectory(w, data, lambda ww, b: step_las(ww, b, num_layers, nl, attention), steps)
    out = dict(meta=np.array([B, T, D, H, num_layers, C, steps, seed, U, K, F]), losses=losses)
    out.update(batch_arrays(data, steps))
    out.update(pack('w:', w))
    out.update(pack('g:', g0))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    return losses


def cfg1_exact_setup():
    """BASELINE.json configs[0]: DBLSTM 2x256 + CTC, batch 8 x 200 x 40 (parity lengths)"""
    B, T, D, H, C = 8, 200, 40, 256, 40
    names, _ = encoder_names('DBLSTM', 2, D, H)
    names += ctc_decoder_names(2 * H, C)
    data = SyntheticData(B, T, D, min_frames=120, min_labels=10, max_labels=40, time_reduction=1, seed=1234)
    return names, data


def sample_index(name, size, n=32):
    seed = sum(ord(c) for c in name)
    return np.random.default_rng([7, seed]).integers(0, size, n)


def make_cfg1_exact(steps=3):
    names, data = cfg1_exact_setup()
    w = draw_weights(names)
    losses, g0, _ = trajectory(w, data, lambda ww, b: step_ctc(ww, b, 'DBLSTM', 2), steps)
    out = dict(losses=losses)
    for k, g in g0.items():
        flat = np.asarray(g, np.float64).ravel()
        out['gnorm:' + k.replace('/', '|')] = np.array(np.sqrt((flat ** 2).sum()))
        out['gsample:' + k.replace('/', '|')] = flat[sample_index(k, flat.size)]
    np.savez_compressed(os.path.join(HERE, 'cfg1_exact.npz'), **out)
    return losses


# ---------------------------------------------------------------- BASELINE configs at their FULL size
# (name -> setup): inputs and weights are regenerated from seeds on both sides, the fixture holds
# the expected losses of `steps` clip+Adam steps, and for step 0 every variable's gradient norm
# plus 32 sampled gradient entries.  SURVEY.md 8(d): parity lengths are ragged (at least one
# full-length utterance), seeds 2234 / 3234 / 5234.
def exact_setup(name):
    """-> (names, data, step_fn, loss_name, recipe, steps)"""
    if name == 'cfg2_exact':      # BASELINE.json configs[1]: the headline config, 32 x 1000 x 40
        B, T, D, H, C = 32, 1000, 40, 512, 40
        names, E = encoder_names('Listener', 3, D, H)
        names += ctc_decoder_names(E, C)
        data = SyntheticData(B, T, D, min_frames=600, min_labels=20, max_labels=60, time_reduction=8, seed=2234)
        return names, data, (lambda ww, b: step_ctc(ww, b, 'Listener', 3)), 'CTC', 'cfg2_listener_ctc', 2
    if name == 'cfg3_exact':      # configs[2]: cfg2 encoder + Speller 1x512, Bahdanau attention
        B, T, D, H, C, U = 32, 1000, 40, 512, 40, 512
        names, E = encoder_names('Listener', 3, D, H)
        names += speller_names(E, U, C, 1, 'vanilla')
        data = SyntheticData(B, T, D, min_frames=600, min_labels=20, max_labels=79, eos=True, time_reduction=8,
                             seed=3234)
        return (names, data, (lambda ww, b: step_las(ww, b, 3, 1, 'vanilla')), 'average_cross_entropy',
                'cfg3_las_vanilla', 2)
    if name == 'cfg5_exact':      # configs[4], one GPU's share: 64 x 1600 x 80, location-aware attention
        B, T, D, H, C, U = 64, 1600, 80, 512, 40, 512
        names, E = encoder_names('Listener', 3, D, H)
        names += speller_names(E, U, C, 1, 'location_aware', K=101, F=10)
        data = SyntheticData(B, T, D, min_frames=960, min_labels=40, max_labels=159, eos=True, time_reduction=8,
                             seed=5234)
        return (names, data, (lambda ww, b: step_las(ww, b, 3, 1, 'location_aware')), 'average_cross_entropy',
                'cfg5_las_location', 1)
    raise KeyError(name)


def make_exact(name):
    import time
    names, data, step_fn, _, _, steps = exact_setup(name)
    w = draw_weights(names)
    t0 = time.time()
    losses, g0, _ = trajectory(w, data, step_fn, steps)
    out = dict(losses=losses, oracle_seconds=np.array(time.time() - t0))
    for k, g in g0.items():
        flat = np.asarray(g, np.float64).ravel()
        out['gnorm:' + k.replace('/', '|')] = np.array(np.sqrt((flat ** 2).sum()))
        out['gsample:' + k.replace('/', '|')] = flat[sample_index(k, flat.size)]
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    return losses


def make_components():
    """small known-answer vectors for the individual kernels"""
    rng = np.random.default_rng(5)
    out = {}
    # CTC (blank = C-1), ragged, with a repeated label and an empty label sequence
    B, T, C = 4, 9, 6
    lg = rng.normal(size=(B, T, C))
    ll = np.array([9, 7, 9, 3], np.int32)
    lab = np.array([[1, 1, 2, 0], [3, 0, 0, 0], [0, 4, 4, 2], [0, 0, 0, 0]], np.int32)
    la = np.array([3, 1, 4, 0], np.int32)
    nll, dlg = O.ctc_loss(lg, ll, lab, la)
    out.update(ctc_logits=lg, ctc_len=ll, ctc_labels=lab, ctc_label_len=la, ctc_nll=nll, ctc_grad=dlg)
    # one BLSTM layer, ragged lengths
    B, T, D, H = 3, 7, 5, 4
    x = rng.normal(size=(B, T, D))
    lens = np.array([7, 4, 1], np.int32)
    x *= (np.arange(T)[None, :, None] < lens[:, None, None])
    p = {'%s_%s' % (d, k): rng.normal(size=s) * 0.4 for d in ('fw', 'bw')
         for k, s in (('kernel', (D + H, 4 * H)), ('bias', (4 * H,)))}
    y, cache = O.blstm_fwd(x, lens, p)
    dy = rng.normal(size=y.shape)
    dx, g = O.blstm_bwd(dy, cache)
    out.update(lstm_x=x, lstm_len=lens, lstm_out=y, lstm_dout=dy, lstm_dx=dx)
    out.update({'lstm_p_' + k: v for k, v in p.items()})
    out.update({'lstm_g_' + k: v for k, v in g.items()})
    # TF-style Adam with clipping, 3 updates
    th = rng.normal(size=50)
    m = np.zeros(50)
    v = np.zeros(50)
    gs = rng.normal(size=(3, 50)) * 2.0
    th0 = th.copy()
    for t in range(3):
        th, m, v = O.clip_adam_update(th, gs[t], m, v, t + 1, 1e-2)
    out.update(adam_theta0=th0, adam_grads=gs, adam_theta=th, adam_m=m, adam_v=v)
    np.savez_compressed(os.path.join(HERE, 'components.npz'), **out)


def make_decode():
    """inference, from oracle/decode_oracle.py: CTC beam search at the default
    beam on posteriors of cfg2's logit shape statistics, edit distances, and an attention beam
    search on a small Speller (weights by the reference's variable names)"""
    from oracle import decode_oracle as DO
    rng = np.random.default_rng(4321)
    out = {}
    B, T, C = 6, 40, 40
    logits = rng.normal(0, 1, (B, T, C)).astype(np.float32)
    path = rng.integers(0, C, (B, T))
    logits[np.arange(B)[:, None], np.arange(T)[None, :], path] += 3.0
    lens = np.array([40, 33, 0, 25, 40, 12], np.int32)
    for merge in (1, 0):
        hyps = DO.ctc_decode_batch(logits, lens, 100, bool(merge))
        ids = np.full((B, T), -1, np.int32)
        for b, h in enumerate(hyps):
            ids[b, :len(h)] = h
        out['ctc_ids_merge%d' % merge] = ids
        out['ctc_len_merge%d' % merge] = np.array([len(h) for h in hyps], np.int32)
    refs = rng.integers(0, C - 1, (B, 30)).astype(np.int32)
    ref_len = rng.integers(0, 31, B).astype(np.int32)
    hyps = DO.ctc_decode_batch(logits, lens, 100, True)
    out.update(ctc_logits=logits, ctc_lens=lens, ed_ref=refs, ed_ref_len=ref_len,
               ed_dist=np.array([DO.edit_distance(h, list(refs[b, :ref_len[b]])) for b, h in enumerate(hyps)], np.int32))
    # attention beam search
    Bs, Te, E, U, Cs, W, S = 3, 11, 24, 32, 9, 6, 12
    for attention, K, F in (('vanilla', 0, 0), ('location_aware', 5, 3)):
        names = speller_names(E, U, Cs, 1, attention, K, F)
        w = draw_weights(names)
        w[SP + 'dense/kernel'] = (w[SP + 'dense/kernel'] * 6.0).astype(np.float32)    # well separated hypotheses
        w[SP + 'dense/bias'][Cs - 1] = 1.0                                             # ... some of which end
        enc_len = np.array([11, 6, 9], np.int32)
        enc = rng.normal(size=(Bs, Te, E)).astype(np.float32)
        enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
        res = DO.speller_beam_search(enc.astype(np.float64), enc_len, speller_view(w, 1, attention), W, S, 1.0, 1.0,
                                     attention)
        pre = 'bs_%s_' % attention
        out.update(pack(pre + 'w_', w))
        out.update({pre + 'enc': enc, pre + 'enc_len': enc_len, pre + 'seq': res['sequences'].astype(np.int32),
                    pre + 'len': res['lengths'].astype(np.int32), pre + 'scores': res['scores'].astype(np.float32),
                    pre + 'align': res['alignments'].astype(np.float32)})
    np.savez_compressed(os.path.join(HERE, 'decode.npz'), **out)
    return out['ctc_len_merge1'].tolist()


if __name__ == '__main__':
    if len(sys.argv) > 1:             # full-size fixtures, one at a time (minutes of float64 each):
        for n in sys.argv[1:]:        #   python tests/golden/make_golden.py cfg2_exact cfg3_exact cfg5_exact
            print('%-14s' % n, make_exact(n), flush=True)
        sys.exit(0)
    make_components()
    print('decode        ', make_decode())
    print('cfg1_exact    ', make_cfg1_exact())
    print('cfg1_small    ', make_small_ctc('cfg1_small', 'DBLSTM', 2, 32, 4, 40, 1, 1234))
    print('cfg2_small    ', make_small_ctc('cfg2_small', 'Listener', 3, 32, 4, 64, 8, 2234))
    print('cfg3_small    ', make_small_las('cfg3_small', 'vanilla', 32, 32, 4, 64, 3234))
    print('cfg5_small    ', make_small_las('cfg5_small', 'location_aware', 32, 32, 4, 64, 5234, K=7, F=4))
