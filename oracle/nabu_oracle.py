"""CPU oracle for the Nabu training hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy restatement of the arithmetic that the reference
(vrenkens/nabu, TensorFlow 1.8 graph mode) executes in one training step.  It is
the checker the HIP kernels are compared against.  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; ``nabu_amd`` never does.

PARITY UNPINNED: the reference ships no unit tests, golden vectors or fixtures
for this path and neither Python 2 nor TensorFlow 1.8 can
be run in the build container, so the oracle could not be checked against the
reference itself.  The arithmetic lives in a third-party dependency that is not
under /root/reference: TensorFlow 1.8.0 (README.md:9).  The oracle restates the
published TF-1.8 semantics of the ops the reference calls and is pinned by
(tests/test_oracle.py): independent PyTorch-CPU float64 autograd restatements,
torch.nn.functional.ctc_loss, torch.nn.LSTM, central finite differences and
closed-form known answers.

Reference call sites restated here (paths relative to /root/reference):
  nabu/neuralnetworks/components/layer.py:8-51     blstm  -> blstm_fwd / blstm_bwd
  nabu/neuralnetworks/components/layer.py:53-94    pblstm -> pyramid_stack_*
  nabu/neuralnetworks/components/ops.py:6-60       pyramid_stack
  nabu/neuralnetworks/models/ed_encoders/listener.py:14-74   listener_fwd/bwd
  nabu/neuralnetworks/models/ed_encoders/dblstm.py:11-59     dblstm_fwd/bwd
  nabu/neuralnetworks/models/ed_decoders/dnn_decoder.py:53-57  linear_fwd/bwd
  nabu/neuralnetworks/trainers/loss_functions.py:180-214    ctc_loss
  nabu/neuralnetworks/trainers/loss_functions.py:78-109,155-165  average_cross_entropy
  nabu/neuralnetworks/models/ed_decoders/rnn_decoder.py:13-82,
  nabu/neuralnetworks/models/ed_decoders/speller.py:13-69,
  nabu/neuralnetworks/components/attention.py:6-39,90-292,
  nabu/neuralnetworks/components/rnn_cell.py:109-155        speller_fwd/bwd
  nabu/neuralnetworks/trainers/trainer.py:153-166   learning_rate
  nabu/neuralnetworks/trainers/trainer.py:512-580   clip_adam_update

All functions take a ``dtype`` implicitly from their inputs: float64 for parity
checks, float32 when the oracle is timed as the CPU baseline ("port").
"""

import numpy as np

FORGET_BIAS = 1.0          # tf.contrib.rnn.{LayerNormBasicLSTMCell,LSTMCell} default


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# initialisers (TF-1.8 scope default = glorot_uniform_initializer)
# --------------------------------------------------------------------------
def glorot_uniform(rng, shape, dtype=np.float32):
    """tf.glorot_uniform_initializer: U(-l, l), l = sqrt(6 / (fan_in + fan_out)).

    For rank-1 shapes TF uses fan_in = fan_out = shape[0]; for conv kernels
    [k, in, out] the receptive field multiplies both fans."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)


# --------------------------------------------------------------------------
# LSTM direction  (layer.py:35-47 -> LayerNormBasicLSTMCell(layer_norm=False)
#                  inside bidirectional_dynamic_rnn(sequence_length=...))
# --------------------------------------------------------------------------
def lstm_dir_fwd(x, lens, kernel, bias, reverse):
    """One direction of layer.blstm.

    x [B,T,D], lens [B] int, kernel [(D+H),4H] (rows: input then recurrent,
    gate column blocks i, j, f, o), bias [4H].
    Semantics (TF 1.8): z = [x_t, h]·kernel + bias; c = c·σ(f+1) + σ(i)·tanh(j);
    h = tanh(c)·σ(o); for t >= len the output row is 0 and (c, h) are frozen;
    the backward direction runs over reverse_sequence(x, len), i.e. starts at
    each sequence's own last valid frame.
    Returns out [B,T,H] and a cache for lstm_dir_bwd."""
    B, T, D = x.shape
    H = kernel.shape[1] // 4
    dt = x.dtype
    Wx, Wh = kernel[:D], kernel[D:]
    out = np.zeros((B, T, H), dt)
    gates = np.zeros((B, T, 4 * H), dt)      # post-activation i, g, f, o
    cs = np.zeros((B, T, H), dt)             # c_t
    hprev = np.zeros((B, T, H), dt)          # h_{t-1} as seen by step t
    cprev = np.zeros((B, T, H), dt)
    h = np.zeros((B, H), dt)
    c = np.zeros((B, H), dt)
    lens = np.asarray(lens)
    ar = np.arange(B)
    for s in range(int(lens.max()) if B else 0):
        act = s < lens
        t = np.where(reverse, lens - 1 - s, s)
        t = np.where(act, t, 0)
        xt = x[ar, t]
        z = xt @ Wx + h @ Wh + bias
        i = sigmoid(z[:, :H])
        g = np.tanh(z[:, H:2 * H])
        f = sigmoid(z[:, 2 * H:3 * H] + FORGET_BIAS)
        o = sigmoid(z[:, 3 * H:])
        cn = c * f + i * g
        hn = np.tanh(cn) * o
        a = act[:, None]
        ia = np.nonzero(act)[0]
        ta = t[ia]
        hprev[ia, ta] = h[ia]
        cprev[ia, ta] = c[ia]
        gates[ia, ta] = np.concatenate([i, g, f, o], 1)[ia]
        cs[ia, ta] = cn[ia]
        out[ia, ta] = hn[ia]
        c = np.where(a, cn, c)
        h = np.where(a, hn, h)
    cache = dict(x=x, lens=lens, kernel=kernel, reverse=reverse, gates=gates,
                 cs=cs, hprev=hprev, cprev=cprev)
    return out, cache


def lstm_dir_bwd(dout, cache):
    """Gradient of lstm_dir_fwd.  dout [B,T,H] (rows past len are ignored).
    Returns dx [B,T,D], dkernel, dbias."""
    x, lens, kernel, reverse = cache['x'], cache['lens'], cache['kernel'], cache['reverse']
    gates, cs, hprev, cprev = cache['gates'], cache['cs'], cache['hprev'], cache['cprev']
    B, T, D = x.shape
    H = kernel.shape[1] // 4
    dt = x.dtype
    Wx, Wh = kernel[:D], kernel[D:]
    dx = np.zeros_like(x)
    dz_all = np.zeros((B, T, 4 * H), dt)
    dh = np.zeros((B, H), dt)
    dc = np.zeros((B, H), dt)
    ar = np.arange(B)
    for s in range(int(lens.max()) - 1, -1, -1):
        act = s < lens
        t = np.where(reverse, lens - 1 - s, s)
        t = np.where(act, t, 0)
        g4 = gates[ar, t]
        i, g, f, o = g4[:, :H], g4[:, H:2 * H], g4[:, 2 * H:3 * H], g4[:, 3 * H:]
        tc = np.tanh(cs[ar, t])
        dht = dout[ar, t] + dh
        do = dht * tc
        dct = dc + dht * o * (1 - tc * tc)
        dz = np.concatenate([dct * g * i * (1 - i),
                             dct * i * (1 - g * g),
                             dct * cprev[ar, t] * f * (1 - f),
                             do * o * (1 - o)], 1)
        a = act[:, None]
        dz = np.where(a, dz, 0)
        ia = np.nonzero(act)[0]
        dz_all[ia, t[ia]] = dz[ia]
        dx[ia, t[ia]] = (dz @ Wx.T)[ia]
        dh = np.where(a, dz @ Wh.T, dh)
        dc = np.where(a, dct * f, dc)
    xin = np.concatenate([x, hprev], 2).reshape(B * T, D + H)
    dkernel = xin.T @ dz_all.reshape(B * T, 4 * H)
    dbias = dz_all.sum((0, 1))
    return dx, dkernel, dbias


def blstm_fwd(x, lens, p):
    """layer.blstm (layer.py:8-51): concat(fw, bw) on the feature axis.
    p = dict(fw_kernel, fw_bias, bw_kernel, bw_bias)."""
    of, cf = lstm_dir_fwd(x, lens, p['fw_kernel'], p['fw_bias'], False)
    ob, cb = lstm_dir_fwd(x, lens, p['bw_kernel'], p['bw_bias'], True)
    return np.concatenate([of, ob], 2), (cf, cb)


def blstm_bwd(dout, cache):
    cf, cb = cache
    H = dout.shape[2] // 2
    dxf, dkf, dbf = lstm_dir_bwd(dout[:, :, :H], cf)
    dxb, dkb, dbb = lstm_dir_bwd(dout[:, :, H:], cb)
    return dxf + dxb, dict(fw_kernel=dkf, fw_bias=dbf, bw_kernel=dkb, bw_bias=dbb)


# --------------------------------------------------------------------------
# pyramid_stack (ops.py:6-60)
# --------------------------------------------------------------------------
def pyramid_stack_fwd(x, lens, numsteps):
    """out[b,t',:] = concat_k x[b, numsteps*t'+k, :], T zero-padded to a multiple
    (ops.py:32-38); len' = ceil(len/numsteps) (ops.py:56-58)."""
    B, T, F = x.shape
    Tp = -(-T // numsteps) * numsteps
    xp = np.zeros((B, Tp, F), x.dtype)
    xp[:, :T] = x
    out = xp.reshape(B, Tp // numsteps, numsteps * F)
    return out, -(-np.asarray(lens) // numsteps)


def pyramid_stack_bwd(dout, T, numsteps):
    B, Tq, FF = dout.shape
    return dout.reshape(B, Tq * numsteps, FF // numsteps)[:, :T]


# --------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------
def listener_fwd(x, lens, layers, pyramid_steps=2):
    """Listener.encode (listener.py:49-65) with input_noise=0, dropout=1:
    len(layers)-1 pblstm layers followed by one plain blstm."""
    caches = []
    h, l = x, np.asarray(lens)
    for p in layers[:-1]:
        o, c = blstm_fwd(h, l, p)
        T = o.shape[1]
        h, l2 = pyramid_stack_fwd(o, l, pyramid_steps)
        caches.append((c, T))
        l = l2
    o, c = blstm_fwd(h, l, layers[-1])
    caches.append((c, o.shape[1]))
    return o, l, caches


def listener_bwd(dout, caches, pyramid_steps=2):
    grads = []
    c, _ = caches[-1]
    d, g = blstm_bwd(dout, c)
    grads.append(g)
    for c, T in reversed(caches[:-1]):
        d = pyramid_stack_bwd(d, T, pyramid_steps)
        d, g = blstm_bwd(d, c)
        grads.append(g)
    return d, grads[::-1]


def dblstm_fwd(x, lens, layers):
    """DBLSTM.encode (dblstm.py:44-54): stacked blstm at full time resolution."""
    caches = []
    h = x
    for p in layers:
        h, c = blstm_fwd(h, lens, p)
        caches.append(c)
    return h, np.asarray(lens), caches


def dblstm_bwd(dout, caches):
    grads = []
    d = dout
    for c in reversed(caches):
        d, g = blstm_bwd(d, c)
        grads.append(g)
    return d, grads[::-1]


# --------------------------------------------------------------------------
# DNNDecoder outlayer (dnn_decoder.py:53-57, tf.contrib.layers.linear)
# --------------------------------------------------------------------------
def linear_fwd(x, W, b):
    return x @ W + b


def linear_bwd(dout, x, W):
    F = x.shape[-1]
    x2, d2 = x.reshape(-1, F), dout.reshape(-1, dout.shape[-1])
    return (d2 @ W.T).reshape(x.shape), x2.T @ d2, d2.sum(0)


# --------------------------------------------------------------------------
# CTC (loss_functions.py:180-214 -> tf.nn.ctc_loss, time_major=False)
# --------------------------------------------------------------------------
def relu_fwd(x):
    return np.maximum(x, 0)


def relu_bwd(dy, y):
    return np.where(y > 0, dy, 0)


def layer_norm_fwd(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm(x) as called at dnn_decoder.py:46-47 with the TF-1.8
    defaults begin_norm_axis=1, begin_params_axis=-1, variance_epsilon=1e-12: moments over
    every axis but the batch axis (for [B,T,F]: over time AND features, padded frames
    included); gamma/beta [F]."""
    B = x.shape[0]
    flat = x.reshape(B, -1)
    mu = flat.mean(1)
    var = flat.var(1)
    rstd = 1.0 / np.sqrt(var + eps)
    sh = (B,) + (1,) * (x.ndim - 1)
    xh = (x - mu.reshape(sh)) * rstd.reshape(sh)
    return xh * gamma + beta, (xh, rstd, gamma)


def layer_norm_bwd(dy, cache):
    xh, rstd, gamma = cache
    B = xh.shape[0]
    sh = (B,) + (1,) * (xh.ndim - 1)
    g = dy * gamma
    m1 = g.reshape(B, -1).mean(1).reshape(sh)
    m2 = (g * xh).reshape(B, -1).mean(1).reshape(sh)
    dx = rstd.reshape(sh) * (g - m1 - xh * m2)
    red = tuple(range(xh.ndim - 1))
    return dx, (dy * xh).sum(red), dy.sum(red)


def _logsumexp2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid='ignore'):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), -np.inf, r)


def _shift(a, k):
    """r[s] = a[s-k] with -inf shifted in (k may be negative)."""
    r = np.full_like(a, -np.inf)
    n = a.shape[0]
    if k >= 0:
        if k < n:
            r[k:] = a[:n - k]
    elif -k < n:
        r[:n + k] = a[-k:]
    return r


def ctc_loss(logits, logit_len, labels, label_len):
    """tf.nn.ctc_loss semantics (TF 1.8): softmax applied internally,
    blank = C-1, ctc_merge_repeated=True, preprocess_collapse_repeated=False,
    frames >= logit_len ignored (zero gradient); raises if a label sequence has
    no valid alignment (ignore_longer_outputs_than_inputs=False).

    logits [B,T,C]; labels [B,Lmax] int (padding ignored); returns
    nll [B] and dnll/dlogits [B,T,C] (per-utterance gradient; the reference's
    loss is mean_b(nll): loss_functions.py:206-212)."""
    B, T, C = logits.shape
    blank = C - 1
    dt = logits.dtype
    nll = np.zeros(B, dt)
    grad = np.zeros_like(logits)
    for b in range(B):
        Tb, L = int(logit_len[b]), int(label_len[b])
        lab = np.asarray(labels[b][:L], dtype=np.int64)
        if np.any(lab < 0) or np.any(lab >= blank):
            raise ValueError('label out of range')
        rep = int(np.sum(lab[1:] == lab[:-1]))
        if L + rep > Tb:
            raise ValueError('Not enough time for target transition sequence')
        S = 2 * L + 1
        ext = np.full(S, blank, np.int64)
        ext[1::2] = lab
        x = logits[b, :Tb]
        m = x.max(1, keepdims=True)
        y = x - m - np.log(np.exp(x - m).sum(1, keepdims=True))   # log-softmax
        ye = y[:, ext]                                             # [Tb,S]
        skip = np.zeros(S, bool)                                   # s-2 -> s allowed
        skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        ninf = -np.inf
        alpha = np.full((Tb, S), ninf, dt)
        alpha[0, 0] = ye[0, 0]
        if S > 1:
            alpha[0, 1] = ye[0, 1]
        for t in range(1, Tb):
            a = alpha[t - 1]
            a1 = _shift(a, 1)
            a2 = np.where(skip, _shift(a, 2), ninf)
            alpha[t] = _logsumexp2(_logsumexp2(a, a1), a2) + ye[t]
        beta = np.full((Tb, S), ninf, dt)
        beta[Tb - 1, S - 1] = ye[Tb - 1, S - 1]
        if S > 1:
            beta[Tb - 1, S - 2] = ye[Tb - 1, S - 2]
        skipf = np.zeros(S, bool)                                  # s -> s+2 allowed
        skipf[:-2] = skip[2:]
        for t in range(Tb - 2, -1, -1):
            bn = beta[t + 1]
            b1 = _shift(bn, -1)
            b2 = np.where(skipf, _shift(bn, -2), ninf)
            beta[t] = _logsumexp2(_logsumexp2(bn, b1), b2) + ye[t]
        ll = _logsumexp2(alpha[Tb - 1, S - 1], alpha[Tb - 1, S - 2] if S > 1 else ninf)
        nll[b] = -ll
        with np.errstate(invalid='ignore'):
            occ = np.exp(alpha + beta - ye - ll)                   # gamma[t,s]
        occ = np.where(np.isfinite(occ), occ, 0)
        g = np.exp(y)
        for s in range(S):
            g[:, ext[s]] -= occ[:, s]
        grad[b, :Tb] = g
    return nll, grad


# --------------------------------------------------------------------------
# average_cross_entropy (loss_functions.py:155-165, 78-109)
# --------------------------------------------------------------------------
def average_cross_entropy(logits, targets, logit_len, target_len):
    """mean_b( sum_{t<logit_len} xent(logits[b,t], targets[b,t]) / target_len[b] ).
    Returns loss (scalar) and dloss/dlogits."""
    B, L, C = logits.shape
    m = logits.max(2, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(2, keepdims=True))
    logp = logits - lse
    tg = np.asarray(targets)[:, :L].astype(np.int64)
    mask = (np.arange(L)[None, :] < np.asarray(logit_len)[:, None])
    ce = -np.take_along_axis(logp, tg[:, :, None], 2)[:, :, 0] * mask
    tl = np.asarray(target_len).astype(logits.dtype)
    loss = np.mean(ce.sum(1) / tl)
    g = np.exp(logp)
    np.put_along_axis(g, tg[:, :, None], np.take_along_axis(g, tg[:, :, None], 2) - 1, 2)
    g = g * mask[:, :, None] / (tl[:, None, None] * B)
    return loss, g


def sum_cross_entropy(logits, targets, target_len):
    """loss_functions.py:142-153: mean_b( sum_{t<target_len} xent(logits[b,t], targets[b,t]) ) — the
    mask is the target length and nothing is divided."""
    return average_cross_entropy(logits, targets, target_len, np.ones(len(target_len)))


# --------------------------------------------------------------------------
# Speller (RNNDecoder + AttentionWrapper + AttentionProjectionWrapper)
# --------------------------------------------------------------------------
def conv1d_same(a, w):
    """tf.layers.conv1d(padding='same', use_bias=False) on a [B,T] signal with
    one input channel.  w [K, numfilt]; returns [B,T,numfilt].
    out[t,f] = sum_d a[t + d - pad_before] w[d,f], pad_before = (K-1)//2."""
    B, T = a.shape
    K, F = w.shape
    pb = (K - 1) // 2
    ap = np.zeros((B, T + K - 1), a.dtype)
    ap[:, pb:pb + T] = a
    win = np.stack([ap[:, d:d + T] for d in range(K)], 2)          # [B,T,K]
    return win @ w


def conv1d_same_bwd(dout, a, w):
    B, T = a.shape
    K, F = w.shape
    pb = (K - 1) // 2
    ap = np.zeros((B, T + K - 1), a.dtype)
    ap[:, pb:pb + T] = a
    win = np.stack([ap[:, d:d + T] for d in range(K)], 2)
    dw = np.einsum('btk,btf->kf', win, dout)
    dwin = dout @ w.T                                              # [B,T,K]
    dap = np.zeros_like(ap)
    for d in range(K):
        dap[:, d:d + T] += dwin[:, :, d]
    return dap[:, pb:pb + T], dw


def _prob_fwd(score, mask, kind):
    """attention.py:9-13,41-55 + _maybe_mask_score(-inf)."""
    if kind == 'softmax':
        s = np.where(mask, score, -np.inf)
        m = s.max(1, keepdims=True)
        e = np.where(mask, np.exp(s - m), 0)
        return e / e.sum(1, keepdims=True)
    sg = np.where(mask, sigmoid(score), 0)
    if kind == 'sigmoid':
        return sg
    if kind == 'normalized_sigmoid':
        return sg / sg.sum(1, keepdims=True)
    raise ValueError(kind)


def _prob_bwd(dalpha, alpha, score, mask, kind):
    if kind == 'softmax':
        return alpha * (dalpha - (alpha * dalpha).sum(1, keepdims=True))
    sg = np.where(mask, sigmoid(score), 0)
    if kind == 'sigmoid':
        return dalpha * sg * (1 - sg)
    z = sg.sum(1, keepdims=True)
    dsg = dalpha / z - (dalpha * sg).sum(1, keepdims=True) / (z * z)
    return np.where(mask, dsg * sg * (1 - sg), 0)


def attention_window(prev_align, left, right):
    """WindowedAttention.__call__ (attention.py:376-390): boolean [B,Te] window from the previous
    alignments — cumsum > 0.5 shifted left by left+1 (padded True) XOR shifted right by `right`
    (padded False), i.e. frames [m-left-1, m+right) around the median frame m."""
    B, Te = prev_align.shape
    half = np.cumsum(prev_align, 1) > 0.5
    sl = np.ones((B, Te), bool)
    if left + 1 < Te:
        sl[:, :Te - left - 1] = half[:, left + 1:]
    sr = np.zeros((B, Te), bool)
    sr[:, right:] = half[:, :Te - right]
    return np.logical_xor(sl, sr)


def speller_fwd(enc, enc_len, targets, target_len, p, attention='vanilla',
                probability_fn='softmax', dec_inputs=None, window=None):
    """RNNDecoder._decode (rnn_decoder.py:13-82) with Speller.create_cell
    (speller.py:13-69), sample_prob=0, dropout=1.

    dec_inputs [B,L] (optional): the decoder input labels to use instead of
    [SOS, targets[:-1]] — the inputs a ScheduledEmbeddingTrainingHelper run with
    sample_prob > 0 actually fed (samples carry no gradient, so given its inputs
    the computation is the deterministic one below).

    enc [B,Te,E], targets [B,Lmax] int (already containing EOS where the recipe
    uses string_eos), target_len [B].
    p: 'lstm' list of dict(kernel [(in+U),4U], bias [4U]) (tf LSTMCell, gate
       order i,j,f,o, forget_bias 1), 'memory_kernel' [E,U], 'query_kernel'
       [U,U], 'attention_v' [U], optionally 'conv_kernel' [K,F] and
       'conv_proj' [F,U] (location_aware), 'out_kernel' [(U+E),C], 'out_bias'.
    Returns logits [B,L,C] (L = max(target_len)), cache."""
    B, Te, E = enc.shape
    dt = enc.dtype
    C = p['out_bias'].shape[0]
    U = p['attention_v'].shape[0]
    enc_len = np.asarray(enc_len)
    target_len = np.asarray(target_len)
    L = int(target_len.max())
    mask = np.arange(Te)[None, :] < enc_len[:, None]               # [B,Te]
    values = enc * mask[:, :, None]                                # _prepare_memory
    keys = values @ p['memory_kernel']                             # memory_layer
    nl = len(p['lstm'])
    hs = [np.zeros((B, U), dt) for _ in range(nl)]
    cs = [np.zeros((B, U), dt) for _ in range(nl)]
    ctx = np.zeros((B, E), dt)
    align = np.zeros((B, Te), dt)
    if attention == 'windowed':
        align[:, 0] = 1                                            # initial_alignments, attention.py:352-359
    sos = C - 1                                                    # rnn_decoder.py:46-47
    inp_ids = np.concatenate([np.full((B, 1), sos, np.int64),
                              np.asarray(targets)[:, :L].astype(np.int64)], 1)
    if dec_inputs is not None:
        inp_ids = np.asarray(dec_inputs).astype(np.int64)
    logits = np.zeros((B, L, C), dt)
    steps = []
    for t in range(L):
        act = (t < target_len)[:, None]
        onehot = np.zeros((B, C), dt)
        onehot[np.arange(B), inp_ids[:, t]] = 1
        x = np.concatenate([onehot, ctx], 1)
        st = dict(act=act, hs_prev=[h.copy() for h in hs], cs_prev=[c.copy() for c in cs],
                  ctx_prev=ctx, align_prev=align, lstm=[])
        nh, nc = [], []
        for n in range(nl):
            xin = np.concatenate([x, hs[n]], 1)
            z = xin @ p['lstm'][n]['kernel'] + p['lstm'][n]['bias']
            i = sigmoid(z[:, :U]); g = np.tanh(z[:, U:2 * U])
            f = sigmoid(z[:, 2 * U:3 * U] + FORGET_BIAS); o = sigmoid(z[:, 3 * U:])
            c = cs[n] * f + i * g
            h = np.tanh(c) * o
            st['lstm'].append(dict(xin=xin, i=i, g=g, f=f, o=o, c=c))
            nh.append(h); nc.append(c)
            x = h
        query = x
        q = query @ p['query_kernel']
        s = keys + q[:, None, :]
        if attention == 'location_aware':
            cf = conv1d_same(align, p['conv_kernel'])
            s = s + cf @ p['conv_proj']
            st['cf'] = cf
        elif attention == 'windowed':
            wmask = attention_window(align, window[0], window[1])
            st['wmask'] = wmask
        elif attention != 'vanilla':
            raise ValueError(attention)
        th = np.tanh(s)
        score = th @ p['attention_v']
        pmask = mask & wmask if attention == 'windowed' else mask
        st['pmask'] = pmask
        al = _prob_fwd(score, pmask, probability_fn)
        cx = np.einsum('bt,bte->be', al, values)
        lg = np.concatenate([query, cx], 1) @ p['out_kernel'] + p['out_bias']
        st.update(query=query, th=th, score=score, al=al, cx=cx)
        steps.append(st)
        logits[:, t] = np.where(act, lg, 0)
        hs = [np.where(act, a, b_) for a, b_ in zip(nh, hs)]
        cs = [np.where(act, a, b_) for a, b_ in zip(nc, cs)]
        ctx = np.where(act, cx, ctx)
        align = np.where(act, al, align)
    cache = dict(steps=steps, p=p, values=values, keys=keys, mask=mask, enc=enc,
                 attention=attention, probability_fn=probability_fn, U=U, C=C)
    return logits, target_len.copy(), cache


def speller_bwd(dlogits, cache):
    """Gradient of speller_fwd: returns d enc and a dict of parameter grads."""
    p, steps = cache['p'], cache['steps']
    values, keys, mask = cache['values'], cache['keys'], cache['mask']
    att, pf, U, C = cache['attention'], cache['probability_fn'], cache['U'], cache['C']
    B, Te, E = values.shape
    dt = values.dtype
    nl = len(p['lstm'])
    g = {k: np.zeros_like(v) for k, v in p.items() if k != 'lstm'}
    g['lstm'] = [dict(kernel=np.zeros_like(q['kernel']), bias=np.zeros_like(q['bias']))
                 for q in p['lstm']]
    dvalues = np.zeros_like(values)
    dkeys = np.zeros_like(keys)
    dhs = [np.zeros((B, U), dt) for _ in range(nl)]
    dcs = [np.zeros((B, U), dt) for _ in range(nl)]
    dctx = np.zeros((B, E), dt)
    dalign = np.zeros((B, Te), dt)
    for t in range(len(steps) - 1, -1, -1):
        st = steps[t]
        act = st['act']
        dl = np.where(act, dlogits[:, t], 0)
        qc = np.concatenate([st['query'], st['cx']], 1)
        g['out_kernel'] += qc.T @ dl
        g['out_bias'] += dl.sum(0)
        dqc = dl @ p['out_kernel'].T
        dquery = dqc[:, :U]
        dcx = dqc[:, U:] + np.where(act, dctx, 0)
        dal = np.einsum('be,bte->bt', dcx, values) + np.where(act, dalign, 0)
        dvalues += st['al'][:, :, None] * dcx[:, None, :]
        dscore = _prob_bwd(dal, st['al'], st['score'], st['pmask'], pf)
        dscore = np.where(act, dscore, 0)
        g['attention_v'] += np.einsum('bt,btu->u', dscore, st['th'])
        ds = dscore[:, :, None] * p['attention_v'] * (1 - st['th'] ** 2)
        dkeys += ds
        dq = ds.sum(1)
        dalign_prev = np.zeros((B, Te), dt)
        if att == 'location_aware':
            g['conv_proj'] += np.einsum('btf,btu->fu', st['cf'], ds)
            dcf = ds @ p['conv_proj'].T
            dalign_prev, dck = conv1d_same_bwd(dcf, st['align_prev'], p['conv_kernel'])
            g['conv_kernel'] += dck
        g['query_kernel'] += st['query'].T @ dq
        dx = dquery + dq @ p['query_kernel'].T
        ndhs, ndcs = [None] * nl, [None] * nl
        for n in range(nl - 1, -1, -1):
            c_ = st['lstm'][n]
            i, gg, f, o, c = c_['i'], c_['g'], c_['f'], c_['o'], c_['c']
            tc = np.tanh(c)
            dh = np.where(act, dx + dhs[n], 0)
            dc = np.where(act, dcs[n], 0) + dh * o * (1 - tc * tc)
            dz = np.concatenate([dc * gg * i * (1 - i), dc * i * (1 - gg * gg),
                                 dc * st['cs_prev'][n] * f * (1 - f),
                                 dh * tc * o * (1 - o)], 1)
            g['lstm'][n]['kernel'] += c_['xin'].T @ dz
            g['lstm'][n]['bias'] += dz.sum(0)
            dxin = dz @ p['lstm'][n]['kernel'].T
            nin = c_['xin'].shape[1] - U
            ndhs[n] = np.where(act, dxin[:, nin:], dhs[n])
            ndcs[n] = np.where(act, dc * f, dcs[n])
            dx = dxin[:, :nin]
        dhs, dcs = ndhs, ndcs
        dctx = np.where(act, dx[:, C:], dctx)            # cell_in = [onehot, ctx_prev]
        dalign = np.where(act, dalign_prev, dalign)
    g['memory_kernel'] += values.reshape(-1, E).T @ dkeys.reshape(-1, U)
    dvalues += dkeys @ p['memory_kernel'].T
    return dvalues * mask[:, :, None], g


# --------------------------------------------------------------------------
# optimiser (trainer.py:153-166, 512-580)
# --------------------------------------------------------------------------
def learning_rate(initial, decay, global_step, num_steps, fact=1.0):
    """tf.train.exponential_decay (non-staircase) * learning_rate_fact."""
    return initial * decay ** (float(global_step) / float(num_steps)) * fact


def clip_adam_update(theta, grad, m, v, t, lr, clip=1.0, b1=0.9, b2=0.999, eps=1e-8):
    """clip_by_value(g,-1,1) per element (trainer.py:560-563) then TF-1.8
    AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)
    (eps added to the UNcorrected sqrt(v)).  t counts Adam applications from 1.
    Returns new (theta, m, v)."""
    g = np.clip(grad, -clip, clip)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v
