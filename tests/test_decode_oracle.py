"""Pins oracle/decode_oracle.py (inference: CTC prefix beam search, attention beam search, edit
distance) — the reference has no tests for these and TF 1.8 cannot run here, so the oracle is
checked against brute force on cases small enough to enumerate."""
import itertools

import numpy as np
import pytest

from oracle import decode_oracle as D


def _all_label_seqs(C, maxlen, blank):
    labs = [c for c in range(C) if c != blank]
    for n in range(maxlen + 1):
        for s in itertools.product(labs, repeat=n):
            yield list(s)


@pytest.mark.parametrize('seed', range(6))
def test_ctc_beam_search_finds_the_most_probable_labelling(seed):
    """a beam wide enough to hold every prefix is an exact search"""
    rng = np.random.default_rng(seed)
    T, C = 4, 3
    logits = rng.normal(0, 2.0, (T, C))
    got = D.ctc_beam_search(logits, beam_width=64, merge_repeated=False)
    best, best_lp = None, -np.inf
    for s in _all_label_seqs(C, T, C - 1):
        lp = D.ctc_label_prob_bruteforce(logits, s)
        if lp > best_lp:
            best, best_lp = s, lp
    assert got == best


def test_ctc_beam_search_narrow_beam_is_greedy_on_peaky_posteriors():
    rng = np.random.default_rng(3)
    T, C = 30, 6
    path = rng.integers(0, C, T)
    logits = rng.normal(0, 0.1, (T, C))
    logits[np.arange(T), path] += 12.0
    want, prev = [], -1
    for s in path:
        if s != prev and s != C - 1:
            want.append(int(s))
        prev = s
    assert D.ctc_beam_search(logits, beam_width=4, merge_repeated=False) == want


def test_ctc_merge_repeated_collapses_output_labels():
    """tf.nn.ctc_beam_search_decoder's default merge_repeated=True also merges a genuinely
    repeated label ('a', blank, 'a'), which the reference inherits (ctc_decoder.py:64-66)"""
    C = 3
    path = [0, 2, 0, 1, 1, 2]
    logits = np.full((len(path), C), -8.0)
    logits[np.arange(len(path)), path] = 8.0
    assert D.ctc_beam_search(logits, 10, merge_repeated=False) == [0, 0, 1]
    assert D.ctc_beam_search(logits, 10, merge_repeated=True) == [0, 1]


def test_ctc_decode_batch_respects_lengths():
    rng = np.random.default_rng(5)
    logits = rng.normal(0, 1, (3, 7, 4))
    lens = [7, 0, 3]
    out = D.ctc_decode_batch(logits, lens, beam_width=8)
    assert out[1] == []
    assert out[2] == D.ctc_beam_search(logits[2, :3], 8)


@pytest.mark.parametrize('a,b,d', [('kitten', 'sitting', 3), ('', 'abc', 3), ('abc', '', 3), ('abc', 'abc', 0),
                                   ('flaw', 'lawn', 2), ('intention', 'execution', 5)])
def test_edit_distance_known_answers(a, b, d):
    assert D.edit_distance(list(a), list(b)) == d


def _speller_params(rng, C, E, U, nl=1, location=False, K=3, F=2):
    p = dict(lstm=[dict(kernel=rng.normal(0, 0.8, ((C + E if n == 0 else U) + U, 4 * U)),
                        bias=rng.normal(0, 0.3, 4 * U)) for n in range(nl)],
             memory_kernel=rng.normal(0, 0.8, (E, U)), query_kernel=rng.normal(0, 0.8, (U, U)),
             attention_v=rng.normal(0, 0.8, U), out_kernel=rng.normal(0, 1.5, (U + E, C)),
             out_bias=rng.normal(0, 0.5, C))
    if location:
        p['conv_kernel'] = rng.normal(0, 0.8, (K, F))
        p['conv_proj'] = rng.normal(0, 0.8, (F, U))
    return p


@pytest.mark.parametrize('attention', ['vanilla', 'location_aware'])
def test_attention_beam_search_is_exhaustive_with_a_wide_beam(attention):
    rng = np.random.default_rng(11)
    B, Te, E, U, C, steps = 2, 5, 4, 4, 3, 3
    p = _speller_params(rng, C, E, U, location=attention == 'location_aware')
    enc = rng.normal(0, 1, (B, Te, E))
    enc_len = np.array([5, 3])
    W = 48                                                  # > number of hypotheses alive at any step
    res = D.speller_beam_search(enc, enc_len, p, W, steps, length_penalty=0.0, attention=attention)
    end = C - 1
    for b in range(B):
        # every hypothesis the search can end with: finished (… EOS) within `steps`, or `steps` labels
        hyps = []
        for n in range(steps + 1):
            for s in itertools.product(range(C - 1), repeat=n):
                if n < steps:
                    hyps.append(list(s) + [end])
                else:
                    hyps.append(list(s))
        scored = sorted(((D.speller_sequence_logprob(enc[b:b + 1], enc_len[b:b + 1], p, h, attention), h)
                         for h in hyps if len(h) <= steps), key=lambda x: -x[0])
        best_lp, best = scored[0]
        L = int(res['lengths'][b, 0])
        got = list(res['sequences'][b, 0, :L])
        assert got == [x for x in best if x != end]
        np.testing.assert_allclose(res['logprobs'][b, 0], best_lp, atol=1e-9)
        # the beam is sorted by score
        assert np.all(np.diff(res['scores'][b]) <= 1e-12)


def test_attention_beam_search_length_penalty_and_stop():
    """finished hypotheses keep their log-probability and length; with the sticky `finished` of
    dynamic_decode the search stops once every beam slot has been finished at some step"""
    rng = np.random.default_rng(2)
    B, Te, E, U, C = 1, 4, 4, 4, 4
    p = _speller_params(rng, C, E, U)
    p['out_bias'][C - 1] += 3.0                              # EOS likely: the search ends early
    enc = rng.normal(0, 1, (B, Te, E))
    res = D.speller_beam_search(enc, [4], p, 3, 50, length_penalty=1.0)
    assert res['sequences'].shape[2] < 50
    lp = ((5.0 + res['lengths']) / 6.0) ** 1.0
    np.testing.assert_allclose(res['scores'], res['logprobs'] / lp, rtol=1e-6)
