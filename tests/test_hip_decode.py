"""GPU parity of the inference decoders (decode.hip through the C ABI) against
oracle/decode_oracle.py: CTC prefix beam search, edit distance, the pruning/gather steps of the
attention beam search and the whole search on a Speller; then the decoders and the
DecoderEvaluator through the recipe API."""
import configparser

import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from nabu_amd import ops, recipes

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ctc_case(rng, B, T, C, peaky):
    logits = rng.normal(0, 1.0, (B, T, C)).astype(np.float32)
    if peaky:
        path = rng.integers(0, C, (B, T))
        logits[np.arange(B)[:, None], np.arange(T)[None, :], path] += peaky
    lens = rng.integers(T // 2, T + 1, B).astype(np.int32)
    lens[0] = T
    return logits, lens


@pytest.mark.parametrize('B,T,C,W,peaky,merge', [
    (4, 12, 5, 8, 0.0, True), (4, 12, 5, 8, 0.0, False), (6, 40, 40, 100, 3.0, True),
    (3, 60, 12, 16, 2.0, True), (2, 25, 3, 100, 0.0, False), (5, 33, 40, 100, 0.5, True)])
def test_ctc_beam_search_matches_oracle(B, T, C, W, peaky, merge):
    rng = np.random.default_rng(B * 100 + T + C)
    logits, lens = _ctc_case(rng, B, T, C, peaky)
    if B > 2:
        lens[1] = 0                                       # an empty utterance decodes to nothing
    ids, out_len, lp = ops.ctc_beam_search(torch.tensor(logits, device=DEV), torch.tensor(lens, device=DEV), W, merge)
    ids, out_len = ids.cpu().numpy(), out_len.cpu().numpy()
    want = D.ctc_decode_batch(logits, lens, W, merge)
    for b in range(B):
        assert list(ids[b, :out_len[b]]) == want[b], b
        assert np.all(ids[b, out_len[b]:] == -1)
    assert np.all(np.isfinite(lp.cpu().numpy()))


def test_ctc_beam_search_cfg2_shape_and_exhaustive_small():
    """BASELINE cfg2's logit shape [32,125,40] at the default beam; and a beam wide enough to be
    exact agrees with brute force over all labellings"""
    rng = np.random.default_rng(7)
    logits, lens = _ctc_case(rng, 32, 125, 40, 4.0)
    ids, out_len, _ = ops.ctc_beam_search(torch.tensor(logits, device=DEV), torch.tensor(lens, device=DEV))
    want = D.ctc_decode_batch(logits[:4], lens[:4], 100, True)
    for b in range(4):
        assert list(ids[b, :out_len[b]].cpu().numpy()) == want[b]
    T, C = 4, 3
    small = rng.normal(0, 2, (8, T, C)).astype(np.float32)
    ids, out_len, lp = ops.ctc_beam_search(torch.tensor(small, device=DEV),
                                           torch.full((8,), T, dtype=torch.int32, device=DEV), 64, False)
    import itertools
    for b in range(8):
        best, best_lp = None, -np.inf
        for n in range(T + 1):
            for s in itertools.product(range(C - 1), repeat=n):
                v = D.ctc_label_prob_bruteforce(small[b].astype(np.float64), list(s))
                if v > best_lp:
                    best, best_lp = list(s), v
        assert list(ids[b, :out_len[b]].cpu().numpy()) == best
        assert abs(float(lp[b]) - best_lp) < 1e-4


def test_edit_distance_matches_oracle():
    rng = np.random.default_rng(1)
    B, Lh, Lt = 40, 37, 50
    hyp = rng.integers(0, 6, (B, Lh)).astype(np.int32)
    ref = rng.integers(0, 6, (B, Lt)).astype(np.int32)
    hl = rng.integers(0, Lh + 1, B).astype(np.int32)
    tl = rng.integers(0, Lt + 1, B).astype(np.int32)
    hl[0], tl[0] = 0, 0
    hl[1], tl[1] = Lh, 0
    hl[2], tl[2] = 0, Lt
    ref[3, :Lh] = hyp[3]; hl[3], tl[3] = Lh, Lh                      # identical
    got = ops.edit_distance(torch.tensor(hyp, device=DEV), torch.tensor(hl, device=DEV),
                            torch.tensor(ref, device=DEV), torch.tensor(tl, device=DEV)).cpu().numpy()
    want = [D.edit_distance(list(hyp[b, :hl[b]]), list(ref[b, :tl[b]])) for b in range(B)]
    np.testing.assert_array_equal(got, want)
    # a long pair (beyond one wavefront of 256 cells per diagonal)
    h = rng.integers(0, 4, (1, 700)).astype(np.int32)
    r = rng.integers(0, 4, (1, 650)).astype(np.int32)
    got = ops.edit_distance(torch.tensor(h, device=DEV), torch.tensor([700], dtype=torch.int32, device=DEV),
                            torch.tensor(r, device=DEV), torch.tensor([650], dtype=torch.int32, device=DEV))
    assert int(got[0]) == D.edit_distance(list(h[0]), list(r[0]))


@pytest.mark.parametrize('W,C,lpw,temp', [(4, 6, 0.0, 1.0), (16, 40, 1.0, 1.0), (8, 5, 0.7, 2.0)])
def test_beam_prune_and_gather(W, C, lpw, temp):
    """one expand+prune step against a direct numpy restatement of beam_search_decoder.py:233-318"""
    rng = np.random.default_rng(W + C)
    B = 3
    logits = rng.normal(0, 2, (B, W, C)).astype(np.float32)
    logprobs = -rng.uniform(0, 5, (B, W)).astype(np.float32)
    lengths = rng.integers(0, 7, (B, W)).astype(np.int32)
    finished = (rng.uniform(size=(B, W)) < 0.3).astype(np.int32)
    seen = finished.copy()
    t = lambda a: torch.tensor(a, device=DEV)
    lp_d, len_d, fin_d, seen_d = t(logprobs), t(lengths), t(finished), t(seen)
    pred, parent, stay, all_seen = ops.beam_prune(t(logits), lp_d, len_d, fin_d, seen_d, temp, lpw)
    # numpy restatement (float32 like TF)
    FMAX = np.finfo(np.float32).max
    x = logits / np.float32(temp)
    nlp = (x - x.max(-1, keepdims=True))
    nlp = nlp - np.log(np.exp(nlp).sum(-1, keepdims=True))
    nlp = np.where(finished[:, :, None] > 0, -FMAX, nlp).astype(np.float32)
    cand_lp = (logprobs[:, :, None] + nlp).reshape(B, W * C)
    cand_ids = np.tile(np.arange(C), (B, W))
    cand_len = np.repeat(lengths, C, 1) + (cand_ids != C - 1)
    all_lp = np.concatenate([cand_lp, np.where(finished > 0, logprobs, -FMAX)], 1).astype(np.float32)
    all_ids = np.concatenate([cand_ids, np.full((B, W), C - 1)], 1)
    all_len = np.concatenate([cand_len, lengths], 1)
    with np.errstate(over='ignore'):
        scores = (all_lp / D._length_penalty(all_len, lpw)).astype(np.float32)
    order = np.argsort(-scores, 1, kind='stable')[:, :W]
    bi = np.arange(B)[:, None]
    np.testing.assert_array_equal(pred.cpu().numpy(), all_ids[bi, order])
    np.testing.assert_array_equal(len_d.cpu().numpy(), all_len[bi, order])
    np.testing.assert_allclose(lp_d.cpu().numpy(), all_lp[bi, order], rtol=1e-5, atol=1e-5)
    st = order >= W * C
    np.testing.assert_array_equal(stay.cpu().numpy(), st.astype(np.int32))
    np.testing.assert_array_equal(parent.cpu().numpy(), np.where(st, order - W * C, order // C))
    np.testing.assert_array_equal(fin_d.cpu().numpy(), (all_ids[bi, order] == C - 1).astype(np.int32))
    np.testing.assert_array_equal(seen_d.cpu().numpy(), seen | (all_ids[bi, order] == C - 1))
    np.testing.assert_array_equal(all_seen.cpu().numpy(), seen_d.cpu().numpy().all(1).astype(np.int32))
    F = 37
    fresh = rng.normal(size=(B, W, F)).astype(np.float32)
    old = rng.normal(size=(B, W, F)).astype(np.float32)
    got = ops.beam_gather(t(fresh), t(old), parent, stay).cpu().numpy()
    par = parent.cpu().numpy()
    want = np.where(st[:, :, None], old[bi, par], fresh[bi, par])
    np.testing.assert_array_equal(got, want)


def _speller(attention, nl, U, C, E, K=5, F=3, seed=5, prob_fn='softmax'):
    from nabu_amd import variables as vs
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory
    over = {'decoder.num_layers': nl, 'decoder.num_units': U, 'decoder.attention': attention,
            'decoder.probability_fn': prob_fn}
    if attention == 'location_aware':
        over.update({'decoder.numfilt': F, 'decoder.filtersize': K})
    if attention == 'windowed':
        over.update({'decoder.left_window_width': 1, 'decoder.right_window_width': 3})
    mc, _, _ = recipes.load_recipe('cfg3_las_vanilla', **over)
    dec = ed_decoder_factory.factory('speller')(mc, {'text': C}, None)
    return dec, vs.VariableStore(seed=seed)


@pytest.mark.parametrize('attention,nl,U,W,lpw,temp', [
    ('vanilla', 1, 32, 4, 0.0, 1.0), ('vanilla', 2, 16, 8, 1.0, 1.0),
    ('location_aware', 1, 32, 6, 1.0, 1.5), ('vanilla', 1, 32, 16, 1.0, 1.0), ('windowed', 1, 32, 5, 1.0, 1.0),
    ('vanilla:normalized_sigmoid', 1, 32, 5, 1.0, 1.0),
    ('vanilla@40', 1, 32, 4, 1.0, 1.0), ('location_aware@40', 1, 32, 3, 0.0, 1.0)])   # 40 frames: sliced attention
def test_speller_beam_search_matches_oracle(attention, nl, U, W, lpw, temp):
    from nabu_amd import variables as vs
    from nabu_amd.autodiff import SeqLen
    from nabu_amd.neuralnetworks.models.ed_decoders import rnn_decoder
    from tests.test_hip_speller import speller_params
    attention, _, frames = attention.partition('@')
    attention, _, prob_fn = attention.partition(':')
    prob_fn = prob_fn or 'softmax'
    rng = np.random.default_rng(U + W)
    B, Te, E, C, S = 3, int(frames or 11), 24, 9, 12
    dec, store = _speller(attention, nl, U, C, E, prob_fn=prob_fn)
    enc_len = np.array([11, 6, 9], np.int32) if Te == 11 else np.array([Te, Te // 2, Te - 7], np.int32)
    enc = rng.normal(size=(B, Te, E)).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    enc_d = torch.tensor(enc, device=DEV)
    with torch.no_grad(), vs.as_default(store), vs.variable_scope(dec.scope):
        cell = dec.create_cell({'features': enc_d}, {'features': SeqLen(enc_len, DEV)}, False)
        # sharpen the output layer so that hypotheses are well separated and EOS happens
        rnn_decoder.cell_parameters(cell, E)
        store.vars['Speller/decoder/dense/kernel'].data.mul_(6.0)
        seqs, lengths, scores, aligns = rnn_decoder.beam_search(cell, enc_d, SeqLen(enc_len, DEV), W, S, lpw, temp)
    p = speller_params(store.state_dict(), nl, attention)
    ref = D.speller_beam_search(enc.astype(np.float64), enc_len, p, W, S, lpw, temp, attention,
                                window=(1, 3) if attention == 'windowed' else None, probability_fn=prob_fn)
    seqs, lengths, scores, aligns = (x.cpu().numpy() for x in (seqs, lengths, scores, aligns))
    assert seqs.shape[2] == ref['sequences'].shape[2]
    live = np.isfinite(ref['scores']) & (ref['scores'] > -1e30)
    assert live[:, 0].all()
    np.testing.assert_array_equal(lengths[live], ref['lengths'][live])
    np.testing.assert_allclose(scores[live], ref['scores'][live], rtol=2e-4, atol=2e-4)
    for b in range(B):
        for w in range(W):
            if live[b, w]:
                L = lengths[b, w]
                np.testing.assert_array_equal(seqs[b, w, :L], ref['sequences'][b, w, :L])
                # every step of the hypothesis' history, finished steps included
                np.testing.assert_array_equal(seqs[b, w], ref['sequences'][b, w])
                np.testing.assert_allclose(aligns[b, w], ref['alignments'][b, w], atol=2e-5)


def _evaluator_conf(decoder, extra):
    conf = configparser.ConfigParser()
    conf.read_dict({'evaluator': {'evaluator': 'decoder_evaluator', 'batch_size': '4', 'numbatches': '2',
                                  'targets': 'text'},
                    'decoder': dict({'decoder': decoder}, **extra)})
    return conf


def test_ctc_decoder_and_decoder_evaluator_through_the_recipe_api(tmp_path):
    """cfg1 model: CTCDecoder output == oracle beam search on the model's logits; the evaluator's
    running error rate == sum of oracle edit distances / number of reference labels"""
    from nabu_amd.autodiff import SeqLen
    from nabu_amd.neuralnetworks.evaluators import evaluator_factory
    from nabu_amd.neuralnetworks.models.model import Model
    from nabu_amd.processing.synthetic import SyntheticData
    mc, tc, _ = recipes.load_recipe('cfg1_dblstm_ctc', **{'encoder.num_units': 32})
    model = Model(mc, int(tc.get('trainer', 'trainlabels')), None, seed=4)
    data = SyntheticData(4, 50, 40, min_frames=30, min_labels=3, max_labels=9, seed=77)
    alphabet = ' '.join('s%d' % i for i in range(39))
    conf = _evaluator_conf('ctc_decoder', {'text_alphabet': alphabet})
    ev = evaluator_factory.factory('decoder_evaluator')(conf, data, model)
    loss, update, nb = ev.evaluate()
    assert nb == 2
    errors = targets = 0
    for i in range(nb):
        update(i)
        batch = ev.data.batch(i)
        x = torch.tensor(batch['inputs']['features'], device=DEV)
        il = SeqLen(batch['input_seq_length']['features'], DEV)
        with torch.no_grad():
            logits, ll = model({'features': x}, {'features': il}, [], [], False)
        lg = logits['text'].cpu().numpy()
        hyps = D.ctc_decode_batch(lg, ll['text'].host, 100, True)
        out = ev.decoder({'features': x}, {'features': il})
        ids, lens = (a.cpu().numpy() for a in out['text'])
        for b, h in enumerate(hyps):
            assert list(ids[b, :lens[b]]) == h
            tl = batch['target_seq_length']['text'][b]
            errors += D.edit_distance(h, list(batch['targets']['text'][b, :tl]))
            targets += tl
    assert abs(loss[0] - errors / targets) < 1e-9
    ev.decoder.write(out, str(tmp_path), ['utt%d' % b for b in range(4)])
    lines = open(tmp_path / 'text').read().strip().split('\n')
    assert len(lines) == 4 and lines[0].startswith('utt0')
    if lens[0]:
        assert lines[0].split(' ')[1:] == ['s%d' % j for j in ids[0, :lens[0]]]


def test_beam_search_decoder_and_evaluator_through_the_recipe_api(tmp_path):
    from nabu_amd.neuralnetworks.evaluators import evaluator_factory
    from nabu_amd.neuralnetworks.models.model import Model
    from nabu_amd.processing.synthetic import SyntheticData
    mc, tc, _ = recipes.load_recipe('cfg3_las_vanilla', **{'encoder.num_units': 32, 'decoder.num_units': 32})
    model = Model(mc, int(tc.get('trainer', 'trainlabels')), None, seed=4)
    data = SyntheticData(4, 64, 40, min_frames=40, min_labels=2, max_labels=6, eos=True, time_reduction=8, seed=78)
    alphabet = ' '.join('s%d' % i for i in range(39))
    conf = _evaluator_conf('beam_search_decoder', {'alphabet': alphabet, 'max_steps': '10', 'beam_width': '4'})
    ev = evaluator_factory.factory('decoder_evaluator')(conf, data, model)
    loss, update, nb = ev.evaluate()
    errors = targets = 0
    for i in range(nb):
        update(i)
        batch = ev.data.batch(i)
        x = torch.tensor(batch['inputs']['features'], device=DEV)
        out = ev.decoder({'features': x}, {'features': batch['input_seq_length']['features']})
        seqs, lens, scores, aligns = out['text']
        assert seqs.shape[:2] == (4, 4) and aligns.shape[:3] == seqs.shape
        assert torch.all(scores[:, :-1] >= scores[:, 1:])
        for b in range(4):
            tl = batch['target_seq_length']['text'][b]
            hyp = list(seqs[b, 0, :lens[b, 0]].cpu().numpy())
            errors += D.edit_distance(hyp, list(batch['targets']['text'][b, :tl - 1]))
            targets += tl
    assert abs(loss[0] - errors / targets) < 1e-9
    ev.decoder.write(out, str(tmp_path), ['u%d' % b for b in range(4)])
    assert len(open(tmp_path / 'u0').read().strip().split('\n')) == 4
    assert np.load(tmp_path / 'u0_alignments.npy').shape[0] == 4


def test_train_then_test_and_decode_scripts(tmp_path):
    """the reference's run train -> run test -> run decode cycle on an experiment directory with
    TFRecord data sets: the trained variables come back from model/network.ckpt.npz, `test` writes
    the label error rate of the CTC decoder to <expdir>/result and `decode` one line per utterance"""
    import os
    from tests.test_data_path import make_dataset
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    from nabu_amd.scripts import test as test_script, decode as decode_script
    expdir = str(tmp_path / 'exp')
    os.makedirs(expdir)
    conf, feats, texts, alphabet = make_dataset(str(tmp_path / 'train'), n=16, dim=40, min_frames=14)
    tst, tfeats, ttexts, _ = make_dataset(str(tmp_path / 'test'), n=7, dim=40, seed=9, min_frames=14)
    conf.read_dict({'testfbank': dict(tst.items('trainfbank')), 'testtext': dict(tst.items('traintext'))})
    mc, tc, ec = recipes.load_recipe('cfg1_dblstm_ctc', **{
        'encoder.num_units': 16, 'trainer.batch_size': 4, 'trainer.num_epochs': 2, 'io.output_dims': 4})
    tc.set('trainer', 'features', 'trainfbank')
    tc.set('trainer', 'targets', 'text')
    tc.set('trainer', 'text', 'traintext')
    tc.set('trainer', 'valid_frequency', '1000')
    ec.read_dict({'evaluator': {'evaluator': 'None'}})
    for name, c in (('database.conf', conf), ('model.cfg', mc), ('trainer.cfg', tc)):
        with open(os.path.join(expdir, name), 'w') as fid:
            c.write(fid)
    te = configparser.ConfigParser()
    te.read_dict({'evaluator': {'evaluator': 'decoder_evaluator', 'batch_size': '3', 'features': 'testfbank',
                                'targets': 'text', 'text': 'testtext'},
                  'decoder': {'decoder': 'ctc_decoder', 'text_alphabet': ' '.join(alphabet)}})
    rc = configparser.ConfigParser()
    rc.read_dict({'recognizer': {'batch_size': '3', 'features': 'testfbank'},
                  'decoder': {'decoder': 'ctc_decoder', 'text_alphabet': ' '.join(alphabet)}})
    for name, c in (('test_evaluator.cfg', te), ('recognizer.cfg', rc)):
        with open(os.path.join(expdir, name), 'w') as fid:
            c.write(fid)
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=conf, modelconf=mc, evaluatorconf=ec,
                                             expdir=expdir, server=None, task_index=0)
    tr.train()
    trained = tr.model.store.state_dict()
    # run test
    ler = test_script.test(expdir)
    assert 0.0 <= ler and float(open(os.path.join(expdir, 'result')).read()) == ler
    model = test_script.load_model(expdir)
    x = torch.zeros((1, 20, 40), device=DEV)
    with torch.no_grad():
        model({'features': x}, {'features': np.array([20], np.int32)}, [], [], False)
    loaded = model.store.state_dict()
    assert set(loaded) == set(trained) and not model.store.restore
    for k in trained:
        np.testing.assert_array_equal(loaded[k], trained[k])
    # the error rate over the first 6 of 7 test utterances (7 // 3 batches of 3), recomputed with the oracle
    from nabu_amd.autodiff import SeqLen
    errors = targets = 0
    names = sorted(tfeats)
    rd = te  # noqa
    from nabu_amd.processing import input_pipeline
    src = input_pipeline.from_sections(conf, ['features'], [['testfbank']], ['text'], [['testtext']],
                                       batch_size=3, numbuckets=1, shuffle=False)
    for i in range(2):
        b = src.batch(i)
        with torch.no_grad():
            lg, ll = model({'features': torch.tensor(b['inputs']['features'], device=DEV)},
                           {'features': SeqLen(b['input_seq_length']['features'], DEV)}, [], [], False)
        hyps = D.ctc_decode_batch(lg['text'].cpu().numpy(), ll['text'].host, 100, True)
        for j, h in enumerate(hyps):
            tl = b['target_seq_length']['text'][j]
            errors += D.edit_distance(h, list(b['targets']['text'][j, :tl]))
            targets += tl
    assert abs(ler - errors / targets) < 1e-9
    # run decode: every test utterance, the last batch smaller
    out = decode_script.decode(expdir)
    lines = open(os.path.join(out, 'text')).read().strip('\n').split('\n')
    assert sorted(l.split(' ')[0] for l in lines) == names
    for l in lines:
        assert all(s in alphabet for s in l.split(' ')[1:] if s)


def test_ctc_beam_search_ties_go_to_the_earlier_candidate():
    """uniform posteriors: every expansion of a frame ties exactly; the beam keeps the earlier
    candidates (existing leaves before expansions, lower leaf, lower label), like a bounded top-N
    container with a strict admission test — the radix select + lowest-index tie rule"""
    for T, C, W in ((6, 4, 8), (9, 6, 20), (5, 40, 100)):
        logits = np.zeros((2, T, C), np.float32)
        logits[1] = 0.5                                          # any constant: still uniform
        lens = np.array([T, T - 1], np.int32)
        for merge in (True, False):
            ids, out_len, _ = ops.ctc_beam_search(torch.tensor(logits, device=DEV), torch.tensor(lens, device=DEV),
                                                  W, merge)
            want = D.ctc_decode_batch(logits, lens, W, merge)
            for b in range(2):
                assert list(ids[b, :out_len[b]].cpu().numpy()) == want[b], (T, C, W, merge, b)
