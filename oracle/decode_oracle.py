"""CPU restatement of the reference's INFERENCE path — TEST INFRASTRUCTURE.

PARITY UNPINNED: like oracle/nabu_oracle.py this file restates algorithms whose arithmetic lives in
TensorFlow 1.8 (absent here: no TF, no Python 2), so nothing below could be checked against an
output of the reference itself.  It follows

  * nabu/neuralnetworks/decoders/ctc_decoder.py:44-68  -> tf.nn.ctc_beam_search_decoder(logits, len)
    with its defaults beam_width=100, top_paths=1, merge_repeated=True.  [TF-1.8 recalled:
    tensorflow/core/util/ctc/ctc_beam_search.h — CTCBeamSearchDecoder::Step/TopPaths,
    ctc_beam_entry.h — BeamEntry::LabelSeq]
  * nabu/neuralnetworks/components/beam_search_decoder.py:68-451 (the reference's own attention
    beam search; this one IS in the reference and is followed line by line), driven by
    tf.contrib.seq2seq.dynamic_decode [TF-1.8 recalled: `finished` is OR-ed over steps because the
    reference's decoder does not set tracks_own_finished]
  * tf.edit_distance(hyp, truth, normalize=False) as used by ctc_decoder.py:112-118 and
    decoders/beam_search_decoder.py:176-179.

What pins it instead: brute-force enumeration of all CTC paths on tiny cases, exhaustive search of
all label sequences for the attention decoder on tiny cases, textbook Levenshtein known answers
(tests/test_decode_oracle.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product path never does.
"""
import heapq
import itertools

import numpy as np

from oracle import nabu_oracle as no

LOG_ZERO = -np.inf


def _lse(a, b):
    if a == LOG_ZERO:
        return b
    if b == LOG_ZERO:
        return a
    m = max(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


# --------------------------------------------------------------------------
# CTC prefix beam search (tf.nn.ctc_beam_search_decoder)
# --------------------------------------------------------------------------
class _Prob(object):
    __slots__ = ('total', 'blank', 'label')

    def __init__(self):
        self.reset()

    def reset(self):
        self.total = self.blank = self.label = LOG_ZERO

    def assign(self, o):
        self.total, self.blank, self.label = o.total, o.blank, o.label


class _Entry(object):
    """ctc_beam_entry.h BeamEntry: a node of the prefix tree"""
    __slots__ = ('parent', 'label', 'children', 'oldp', 'newp', 'order')

    def __init__(self, parent, label):
        self.parent, self.label, self.children = parent, label, None
        self.oldp, self.newp = _Prob(), _Prob()
        self.order = 0

    def active(self):
        return self.newp.total != LOG_ZERO

    def label_seq(self, merge_repeated):
        """BeamEntry::LabelSeq: walk to the root; with merge_repeated consecutive equal labels of
        the OUTPUT collapse (so a genuine 'a a' decodes to 'a' — TF's documented quirk)"""
        labels, prev, c = [], -1, self
        while c.parent is not None:
            if not merge_repeated or c.label != prev:
                labels.append(c.label)
            prev = c.label
            c = c.parent
        return labels[::-1]


def ctc_beam_search(logits, beam_width=100, merge_repeated=True, blank=None):
    """One utterance: logits [T,C] (T = its logit_seq_length).  Returns the label list of the
    most probable leaf.  Ties (equal float totals) are resolved in favour of the earlier
    insertion, as a bounded top-N container with a strict `>` admission test does."""
    T, C = logits.shape
    blank = C - 1 if blank is None else blank
    x = np.asarray(logits, np.float64)
    x = x - x.max(1, keepdims=True)
    x = x - np.log(np.exp(x).sum(1, keepdims=True))        # per-frame constants do not change ranks
    root = _Entry(None, -1)
    root.newp.total = root.newp.blank = 0.0
    leaves = [root]
    for t in range(T):
        inp = x[t]
        branches = sorted(leaves, key=lambda e: -e.newp.total)          # stable: earlier first
        for b in branches:
            b.oldp.assign(b.newp)
        cands = []                                                      # (entry, total) in insertion order
        for b in branches:
            if b.parent is not None:
                if b.parent.active():
                    prev = b.parent.oldp.blank if b.label == b.parent.label else b.parent.oldp.total
                    b.newp.label = _lse(b.newp.label, prev)
                b.newp.label += inp[b.label]
            b.newp.blank = b.oldp.total + inp[blank]
            b.newp.total = _lse(b.newp.blank, b.newp.label)
            cands.append(b)
        fresh = []
        for b in branches:
            if b.oldp.total == LOG_ZERO:
                continue
            if b.children is None:
                b.children = {}
            for c in range(C):
                if c == blank:
                    continue
                ch = b.children.get(c)
                if ch is not None and ch.active():
                    continue                                            # already a leaf, updated above
                if ch is None:
                    ch = b.children[c] = _Entry(b, c)
                prev = b.oldp.blank if c == b.label else b.oldp.total
                ch.newp.blank = LOG_ZERO
                ch.newp.label = inp[c] + prev
                ch.newp.total = ch.newp.label
                if ch.newp.total > LOG_ZERO:
                    cands.append(ch)
                    fresh.append(ch)
                else:
                    ch.newp.reset()
        for i, e in enumerate(cands):
            e.order = i
        keep = heapq.nsmallest(beam_width, cands, key=lambda e: (-e.newp.total, e.order))
        kept = set(id(e) for e in keep)
        for e in cands:
            if id(e) not in kept:
                e.newp.reset()                                          # left the beam: inactive
                e.oldp.reset()
        leaves = keep
    best = min(leaves, key=lambda e: (-e.newp.total,))
    return best.label_seq(merge_repeated)


def ctc_decode_batch(logits, lens, beam_width=100, merge_repeated=True):
    """CTCDecoder.__call__ (ctc_decoder.py:44-68) for a batch: list of label lists"""
    return [ctc_beam_search(np.asarray(logits[b][:int(lens[b])]), beam_width, merge_repeated)
            for b in range(len(lens))]


def ctc_label_prob_bruteforce(logits, labels, blank=None):
    """log P(labels | logits) by enumerating all C^T paths (tiny cases only)"""
    T, C = logits.shape
    blank = C - 1 if blank is None else blank
    lp = logits - np.log(np.exp(logits).sum(1, keepdims=True))
    tot = LOG_ZERO
    for path in itertools.product(range(C), repeat=T):
        col, prev = [], -1
        for s in path:
            if s != prev and s != blank:
                col.append(s)
            prev = s
        if col == list(labels):
            tot = _lse(tot, float(sum(lp[t, s] for t, s in enumerate(path))))
    return tot


# --------------------------------------------------------------------------
# Edit distance
# --------------------------------------------------------------------------
def edit_distance(hyp, truth):
    """tf.edit_distance(normalize=False) of one pair: Levenshtein distance"""
    n, m = len(hyp), len(truth)
    d = list(range(m + 1))
    for i in range(1, n + 1):
        prev, d[0] = d[0], i
        for j in range(1, m + 1):
            cur = d[j]
            d[j] = min(d[j] + 1, d[j - 1] + 1, prev + (hyp[i - 1] != truth[j - 1]))
            prev = cur
    return d[m]


# --------------------------------------------------------------------------
# Attention beam search (components/beam_search_decoder.py)
# --------------------------------------------------------------------------
def speller_cell_step(p, ids, state, values, keys, mask, attention='vanilla', window=None,
                      probability_fn='softmax'):
    """One AttentionProjectionWrapper(AttentionWrapper(MultiRNNCell)) step on N rows
    (speller.py:13-69, rnn_cell.py:145-155) — the body of nabu_oracle.speller_fwd's loop.
    state = (hs, cs, ctx, align); returns logits [N,C], new state."""
    hs, cs, ctx, align = state
    N = ids.shape[0]
    C = p['out_bias'].shape[0]
    U = p['attention_v'].shape[0]
    onehot = np.zeros((N, C), ctx.dtype)
    onehot[np.arange(N), ids] = 1
    x = np.concatenate([onehot, ctx], 1)
    nh, nc = [], []
    for n, lp in enumerate(p['lstm']):
        z = np.concatenate([x, hs[n]], 1) @ lp['kernel'] + lp['bias']
        i = no.sigmoid(z[:, :U]); g = np.tanh(z[:, U:2 * U])
        f = no.sigmoid(z[:, 2 * U:3 * U] + no.FORGET_BIAS); o = no.sigmoid(z[:, 3 * U:])
        c = cs[n] * f + i * g
        h = np.tanh(c) * o
        nh.append(h); nc.append(c)
        x = h
    s = keys + (x @ p['query_kernel'])[:, None, :]
    if attention == 'location_aware':
        s = s + no.conv1d_same(align, p['conv_kernel']) @ p['conv_proj']
    score = np.tanh(s) @ p['attention_v']
    if attention == 'windowed':
        mask = mask & no.attention_window(align, window[0], window[1])
    al = no._prob_fwd(score, mask, probability_fn)
    cx = np.einsum('bt,bte->be', al, values)
    lg = np.concatenate([x, cx], 1) @ p['out_kernel'] + p['out_bias']
    return lg, (nh, nc, cx, al)


def _length_penalty(lengths, w):
    """beam_search_decoder.py:469-483"""
    if w == 0:
        return np.float32(1.0)
    return ((np.float32(5.) + lengths.astype(np.float32)) ** np.float32(w)) / (np.float32(6.) ** np.float32(w))


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def speller_beam_search(enc, enc_len, p, beam_width, max_steps, length_penalty=1.0, temperature=1.0,
                        attention='vanilla', dtype=np.float64, window=None, probability_fn='softmax'):
    """decoders/beam_search_decoder.py:31-114 + components/beam_search_decoder.py:141-451.

    enc [B,Te,E].  Returns dict(sequences [B,W,time] int, lengths [B,W], scores [B,W],
    alignments [B,W,time,Te]); `time` = the number of steps dynamic_decode ran."""
    B, Te, E = enc.shape
    W = int(beam_width)
    C = p['out_bias'].shape[0]
    U = p['attention_v'].shape[0]
    end = C - 1
    FMAX = np.finfo(np.float32).max
    pp = {k: (np.asarray(v, dtype) if not isinstance(v, list) else
              [dict(kernel=np.asarray(l['kernel'], dtype), bias=np.asarray(l['bias'], dtype)) for l in v])
          for k, v in p.items()}
    enc_len = np.asarray(enc_len)
    # tile_batch: each utterance W times in a row (decoders/beam_search_decoder.py:58-66)
    mask = np.repeat(np.arange(Te)[None, :] < enc_len[:, None], W, 0)
    values = np.repeat(np.asarray(enc, dtype), W, 0) * mask[:, :, None]
    keys = values @ pp['memory_kernel']
    N = B * W
    nl = len(pp['lstm'])
    state = ([np.zeros((N, U), dtype) for _ in range(nl)], [np.zeros((N, U), dtype) for _ in range(nl)],
             np.zeros((N, E), dtype), np.zeros((N, Te), dtype))
    if attention == 'windowed':
        state[3][:, 0] = 1
    logprobs = np.concatenate([np.zeros((B, 1)), np.full((B, W - 1), -np.inf)], 1)
    lengths = np.zeros((B, W), np.int64)
    finished = np.zeros((B, W), bool)
    seen_finished = np.zeros((B, W), bool)
    ids = np.full((B, W), end, np.int64)                    # start token = C-1
    pred_hist, parent_hist, align_hist = [], [], []
    bi = np.arange(B)[:, None]

    def rows(a, idx):                                        # a [N,...] viewed [B,W,...], gather beams
        v = a.reshape((B, W) + a.shape[1:])
        return v[bi, idx].reshape(a.shape)

    for time in range(int(max_steps)):
        lg, new_state = speller_cell_step(pp, ids.reshape(N), state, values, keys, mask, attention, window,
                                          probability_fn)
        out = lg.reshape(B, W, C) / temperature
        new_lp = _log_softmax(out)
        new_lp = np.where(finished[:, :, None], -FMAX, new_lp)
        cand_lp = (logprobs[:, :, None] + new_lp).reshape(B, W * C)
        cand_ids = np.tile(np.arange(C), (B, W))
        cand_len = np.repeat(lengths, C, 1)
        cand_len = np.where(cand_ids == end, cand_len, cand_len + 1)
        stay_lp = np.where(finished, logprobs, -FMAX)
        all_lp = np.concatenate([cand_lp, stay_lp], 1)
        all_ids = np.concatenate([cand_ids, np.full((B, W), end)], 1)
        all_len = np.concatenate([cand_len, lengths], 1)
        # float32 like the reference's graph: -FLT_MAX / penalty overflows to -inf for short hypotheses
        with np.errstate(over='ignore'):
            scores = all_lp.astype(np.float32) / _length_penalty(all_len, length_penalty)
        # tf.nn.top_k: descending, equal scores -> lower index first
        order = np.argsort(-scores, 1, kind='stable')[:, :W]
        # reference lines 297-301: parent = idx // C, and where that equals W the slot idx % C.
        # That names the right slot only while beam_width <= output_dim (always so in the recipes:
        # 16 vs 40); the intended slot idx - W*C is used here so that wider beams work too.
        stay = order >= W * C
        parent = np.where(stay, order - W * C, order // C)
        lengths = all_len[bi, order]
        ids = all_ids[bi, order]
        logprobs = all_lp[bi, order]
        # expanded hypotheses take the cell's new state of their parent, "stay" hypotheses the
        # state they had before this step (beam_search_decoder.py:283-285 _concat_states)
        def pick(new, old):
            return np.where(stay.reshape(N)[(slice(None),) + (None,) * (new.ndim - 1)],
                            rows(old, parent), rows(new, parent))
        state = ([pick(a, b_) for a, b_ in zip(new_state[0], state[0])],
                 [pick(a, b_) for a, b_ in zip(new_state[1], state[1])],
                 pick(new_state[2], state[2]), pick(new_state[3], state[3]))
        finished = ids == end
        pred_hist.append(ids.copy()); parent_hist.append(parent.copy())
        align_hist.append(state[3].reshape(B, W, Te).copy())
        seen_finished |= finished                            # dynamic_decode: finished is sticky
        if seen_finished.all():
            break
    Tn = len(pred_hist)
    # finalize: backwards search (beam_search_decoder.py:341-451)
    seqs = np.zeros((B, W, Tn), np.int64)
    aligns = np.zeros((B, W, Tn, Te), dtype)
    beams = np.tile(np.arange(W), (B, 1))
    for t in range(Tn - 1, -1, -1):
        seqs[:, :, t] = pred_hist[t][bi, beams]
        aligns[:, :, t] = align_hist[t][bi, beams]
        beams = parent_hist[t][bi, beams]
    scores = logprobs / _length_penalty(lengths, length_penalty)
    return dict(sequences=seqs, lengths=lengths, scores=scores, alignments=aligns, logprobs=logprobs)


def speller_sequence_logprob(enc, enc_len, p, seq, attention='vanilla'):
    """log P(seq) of ONE utterance under the cell, teacher-forced (tiny-case exhaustive check)"""
    dtype = np.float64
    Te = enc.shape[1]
    mask = np.arange(Te)[None, :] < np.asarray(enc_len)[:, None]
    values = np.asarray(enc, dtype) * mask[:, :, None]
    keys = values @ p['memory_kernel']
    C = p['out_bias'].shape[0]
    U = p['attention_v'].shape[0]
    nl = len(p['lstm'])
    state = ([np.zeros((1, U))] * nl, [np.zeros((1, U))] * nl, np.zeros((1, enc.shape[2])), np.zeros((1, Te)))
    prev, tot = C - 1, 0.0
    for s in seq:
        lg, state = speller_cell_step(p, np.array([prev]), state, values, keys, mask, attention)
        tot += _log_softmax(lg)[0, s]
        prev = s
    return tot
