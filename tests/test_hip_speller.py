"""GPU parity of the Speller (RNNDecoder + LSTMCell + Bahdanau / location-aware
attention + projection) and of the whole LAS training step against the float64
oracle, through the recipe API and the C ABI."""
import numpy as np
import pytest
import torch

from oracle import nabu_oracle as O
from nabu_amd import recipes
from nabu_amd.processing.synthetic import SyntheticData

pytestmark = pytest.mark.gpu

PRE = 'Speller/decoder/'


SCOPES = {'vanilla': 'bahdanau_attention', 'location_aware': 'location_aware_attention',
          'windowed': 'windowed_attention'}


def speller_params(st, nl, attention):
    sc = SCOPES[attention]
    p = dict(memory_kernel=st[PRE + 'memory_layer/kernel'],
             query_kernel=st[PRE + sc + '/query_layer/kernel'],
             attention_v=st[PRE + sc + '/attention_v'],
             out_kernel=st[PRE + 'dense/kernel'], out_bias=st[PRE + 'dense/bias'], lstm=[])
    for n in range(nl):
        q = PRE + 'attention_wrapper/multi_rnn_cell/cell_%d/lstm_cell/' % n
        p['lstm'].append(dict(kernel=st[q + 'kernel'], bias=st[q + 'bias']))
    if attention == 'location_aware':
        ck = st[PRE + sc + '/conv1d/kernel']
        p['conv_kernel'] = ck.reshape(ck.shape[0], ck.shape[2])
        p['conv_proj'] = st[PRE + sc + '/process_conv_features/kernel']

    def f64(x):
        if isinstance(x, dict):
            return {k: f64(v) for k, v in x.items()}
        if isinstance(x, list):
            return [f64(v) for v in x]
        return x.astype(np.float64)
    return f64(p)


def grad_names(nl, attention):
    sc = SCOPES[attention]
    m = {'memory_kernel': PRE + 'memory_layer/kernel', 'query_kernel': PRE + sc + '/query_layer/kernel',
         'attention_v': PRE + sc + '/attention_v', 'out_kernel': PRE + 'dense/kernel',
         'out_bias': PRE + 'dense/bias'}
    if attention == 'location_aware':
        m['conv_kernel'] = PRE + sc + '/conv1d/kernel'
        m['conv_proj'] = PRE + sc + '/process_conv_features/kernel'
    return m


@pytest.mark.parametrize('Te', [13, 70])      # 70 frames: 5 frame slices per utterance in both attention passes
@pytest.mark.parametrize('attention,nl,U,K,F', [
    ('vanilla', 1, 32, 0, 0), ('vanilla', 2, 16, 0, 0),
    ('location_aware', 1, 32, 5, 3), ('location_aware', 2, 16, 4, 2), ('location_aware', 1, 64, 11, 10),
    ('location_aware', 1, 32, 5, 14),           # numfilt > 12: the generic location-aware kernels
    ('windowed', 1, 32, 1, 2), ('windowed', 2, 16, 0, 3),      # K, F = left / right window width
    ('vanilla:sigmoid', 1, 32, 0, 0), ('vanilla:normalized_sigmoid', 2, 16, 0, 0),      # probability_fn
    ('location_aware:normalized_sigmoid', 1, 32, 5, 3), ('location_aware:sigmoid', 1, 32, 5, 14),
    ('windowed:normalized_sigmoid', 1, 32, 1, 3)])
def test_speller_step_matches_oracle(attention, nl, U, K, F, Te):
    """decoder alone on a given 'encoded' tensor: logits, loss and every gradient"""
    enc_len = np.array([13, 9, 13, 4, 7], np.int32) if Te == 13 else np.array([70, 33, 70, 15, 52], np.int32)
    check_speller(attention, nl, U, K, F, enc_len, np.array([6, 3, 5, 6, 1], np.int32))


def test_speller_step_at_the_cfg5_attention_geometry():
    """BASELINE.json configs[4] geometry of the frame-sliced location-aware attention — 64 utterances,
    200 encoder frames (ragged), filtersize 101, numfilt 10 — with the feature widths shrunk (U = 64,
    E = 24) so that the float64 oracle finishes in seconds: forward, loss and every gradient"""
    rng = np.random.default_rng(55)
    enc_len = rng.integers(120, 201, 64).astype(np.int32)
    enc_len[0] = 200
    tlen = rng.integers(1, 6, 64).astype(np.int32)
    tlen[0] = 5
    check_speller('location_aware', 1, 64, 101, 10, enc_len, tlen)
    check_speller('vanilla', 1, 64, 0, 0, enc_len, tlen)


@pytest.mark.parametrize('B,U,E,K,F,Te', [
    (5, 48, 24, 5, 3, 70),        # 3 unit tiles over 8 waves: most waves own none; 5 slices of 14 frames
    (5, 320, 32, 11, 12, 40),     # 20 unit tiles: a third tile only for waves 0-3; 12 filters (the operand's full k range)
    (16, 128, 64, 101, 10, 200),  # the cfg5 attention geometry at U = 128: rows16 products (K = 128 / 512), 25-frame slices
    (7, 256, 48, 4, 1, 9),        # fewer frames than one tile, one filter, even filter width; rows16 with 7 rows
    (3, 512, 16, 7, 2, 33),       # the kernel's widest U; two frame tiles, the second nearly empty
])
def test_step_chain_location_aware_matrix_pipe_kernels(B, U, E, K, F, Te):
    """The matrix-pipe kernels of the location-aware step chain — attn_fwd_loc_mfma_kernel, attn_bwd_loc_mfma_kernel,
    attn_param_grads_mfma_kernel, rows16_kernel (speller.hip, gemm_skinny.hip) — at shapes that reach their edges, against the
    float64 oracle (check_speller: logits, loss and every gradient): the chain is forced (the persistent decoder would take
    most of these shapes); which kernel runs is decided by the shape alone (round 6: the switches that selected the
    kernels' predecessors are gone, the predecessors remain as the general kernels for geometries these do not take)."""
    import os
    rng = np.random.default_rng(B * 1000 + U)
    enc_len = rng.integers(max(Te // 2, 1), Te + 1, B).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(1, 6, B).astype(np.int32)
    tlen[-1] = 5
    os.environ['NABU_SPELLER_PERSIST'] = os.environ['NABU_SPELLER_PERSIST_BWD'] = '0'
    try:
        check_speller('location_aware', 1, U, K, F, enc_len, tlen, E=E)
    finally:
        for k in ('NABU_SPELLER_PERSIST', 'NABU_SPELLER_PERSIST_BWD'):
            os.environ.pop(k, None)


@pytest.mark.parametrize('attention,nl,K,F', [('vanilla', 1, 0, 0), ('location_aware', 1, 5, 3), ('vanilla', 2, 0, 0),
                                              ('windowed', 1, 2, 3)])
def test_speller_step_on_the_fused_and_multi_stream_paths(attention, nl, K, F):
    """shapes that take every round-2 path of the decoder driver — 32 utterances (two sub-batches on two
    streams), E = U = 64 (the [context | h]·kernel product with the LSTM-cell epilogue, the cell's backward pass
    in the epilogue of dq·Wq^T, dz·[Kx^T | Kh^T] as one product; two layers: the unfused fallbacks) — against
    the oracle, and bit-identical with the sub-batching switched off"""
    import os
    rng = np.random.default_rng(77)
    enc_len = rng.integers(20, 41, 32).astype(np.int32)
    enc_len[0] = 40
    tlen = rng.integers(1, 7, 32).astype(np.int32)
    tlen[3] = 6
    got = check_speller(attention, nl, 64, K, F, enc_len, tlen, E=64)
    os.environ['NABU_SPELLER_EPILOGUE'] = '0'
    try:
        ref = check_speller(attention, nl, 64, K, F, enc_len, tlen, E=64)
    finally:
        del os.environ['NABU_SPELLER_EPILOGUE']
    # the epilogue variants reorder float additions (one product over [context | h] instead of two): close, not equal
    assert np.abs(got - ref).max() < 1e-5
    os.environ['NABU_SPELLER_STREAMS'] = '1'       # no sub-batching: the same arithmetic per utterance
    try:
        one = check_speller(attention, nl, 64, K, F, enc_len, tlen, E=64)
    finally:
        del os.environ['NABU_SPELLER_STREAMS']
    np.testing.assert_array_equal(got, one)


def test_persistent_decoder_location_aware_even_filter_width_and_many_filters():
    """'same' padding of an even filter width (left pad (K - 1) // 2), 12 filters (the kernel's register limit), 70
    frames — both persistent passes against the oracle"""
    from nabu_amd.neuralnetworks.models.ed_decoders import rnn_decoder
    rng = np.random.default_rng(9)
    Te = 70
    enc_len = rng.integers(Te // 2, Te + 1, 32).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(1, 7, 32).astype(np.int32)
    tlen[2] = 6
    check_speller('location_aware', 1, 64, 6, 12, enc_len, tlen, E=64)
    assert rnn_decoder.dynamic_decode.last_paths == (1, 1)
    from nabu_amd import ops as hip
    hip.check_persist_status()


@pytest.mark.parametrize('B', [32, 64])
@pytest.mark.parametrize('stream', [0, 1])
def test_persistent_decoder_location_aware(B, stream):
    """the location-aware variants of the persistent kernels — forward: fifth ring with the alignments, conv features,
    feature projection in the score duty; backward: features recomputed, d features handed to the other slices through a
    ring, the carry into the previous step's alignments and the conv kernel's gradient in the kernel, d keys / d v /
    d conv_proj left to attn_param_grads_kernel — a batch of 64 as two launches of 32, and the values slice read from L2
    instead of LDS (what cfg5's geometry needs; forced here): logits and every gradient against the oracle (inside
    check_speller), the logits against the chain"""
    import os
    from nabu_amd.neuralnetworks.models.ed_decoders import rnn_decoder
    rng = np.random.default_rng(21 + B)
    Te = 37
    enc_len = rng.integers(Te // 2, Te + 1, B).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(1, 7, B).astype(np.int32)
    tlen[2] = 6
    os.environ['NABU_SPELLER_STREAM_VALUES'] = str(stream)
    os.environ['NABU_SPELLER_PERSIST_BWD_LOC'] = '2'                   # (by default only with LDS-resident values)
    try:
        got = check_speller('location_aware', 1, 64, 7, 3, enc_len, tlen, E=64)
        assert rnn_decoder.dynamic_decode.last_paths == (1, 1)         # both passes took the persistent launch
        os.environ['NABU_SPELLER_PERSIST'] = '0'
        ref = check_speller('location_aware', 1, 64, 7, 3, enc_len, tlen, E=64)
        assert rnn_decoder.dynamic_decode.last_paths[0] == 0
    finally:
        os.environ.pop('NABU_SPELLER_PERSIST', None)
        del os.environ['NABU_SPELLER_STREAM_VALUES']
        del os.environ['NABU_SPELLER_PERSIST_BWD_LOC']
    assert np.abs(got - ref).max() < 2e-5
    from nabu_amd import ops as hip
    hip.check_persist_status()


@pytest.mark.parametrize('U,E,Te', [(64, 64, 40), (128, 256, 70), (64, 192, 33)])
def test_persistent_decoder_forward(U, E, Te):
    """speller_persist.hip: the step loops of nabu_speller_fwd / _bwd as ONE persistent launch each (B = 32, one layer,
    vanilla softmax attention, teacher forcing) — ragged decoder lengths (frozen rows), ragged encoder lengths (masked
    frames), Te not a multiple of the 8 frame slices, the 64- and 192-register instantiations — logits and every
    gradient against the oracle, and the logits against the step chain (NABU_SPELLER_PERSIST=0) on the same inputs"""
    import os
    rng = np.random.default_rng(5 + U)
    enc_len = rng.integers(Te // 2, Te + 1, 32).astype(np.int32)
    enc_len[0] = Te
    enc_len[5] = 3                       # fewer frames than slices
    tlen = rng.integers(1, 9, 32).astype(np.int32)
    tlen[3] = 8
    got = check_speller('vanilla', 1, U, 0, 0, enc_len, tlen, E=E)
    os.environ['NABU_SPELLER_PERSIST'] = '0'
    try:
        ref = check_speller('vanilla', 1, U, 0, 0, enc_len, tlen, E=E)
    finally:
        del os.environ['NABU_SPELLER_PERSIST']
    assert np.abs(got - ref).max() < 2e-5
    from nabu_amd import ops as hip
    hip.check_persist_status()


def test_persistent_decoder_batch_of_64_runs_as_two_launches():
    """vanilla attention, 64 utterances: forward AND backward as two persistent launches of 32 rows each (row offset,
    whole-batch strides of the time-major tensors) — logits and every gradient against the oracle"""
    from nabu_amd.neuralnetworks.models.ed_decoders import rnn_decoder
    rng = np.random.default_rng(77)
    Te = 40
    enc_len = rng.integers(Te // 2, Te + 1, 64).astype(np.int32)
    enc_len[40] = Te
    tlen = rng.integers(1, 9, 64).astype(np.int32)
    tlen[50] = 8
    check_speller('vanilla', 1, 64, 0, 0, enc_len, tlen, E=64)
    assert rnn_decoder.dynamic_decode.last_paths == (1, 1)
    from nabu_amd import ops as hip
    hip.check_persist_status()


def _decoder_run(over, enc, enc_len, tg, tlen, C, seed):
    """logits, d encoded, every parameter gradient and the decoder inputs actually used, for a decoder built from
    the cfg3 recipe + over; the regularisation RNG starts from `seed`"""
    from nabu_amd import variables as vs
    from nabu_amd.autodiff import Tape, SeqLen, record
    from nabu_amd.neuralnetworks.components import ops as nops
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory, rnn_decoder
    from nabu_amd.neuralnetworks.trainers import loss_functions
    mc, _, _ = recipes.load_recipe('cfg3_las_vanilla', **over)
    dec = ed_decoder_factory.factory('speller')(mc, {'text': C}, None)
    store = vs.VariableStore(seed=3)
    dev = torch.device('cuda')
    nops.set_seed(seed)
    enc_d, src, tgd = torch.tensor(enc, device=dev), torch.tensor(enc, device=dev), torch.tensor(tg, device=dev)
    with vs.as_default(store), Tape() as tape:
        record([src], [enc_d], lambda g: [g])
        logits, lsl, _ = dec({'features': enc_d}, {'features': SeqLen(enc_len, dev)}, {'text': tgd},
                             {'text': SeqLen(tlen, dev)}, True)
        loss = loss_functions.average_cross_entropy({'text': tgd}, logits, lsl, {'text': SeqLen(tlen, dev)})
    used = rnn_decoder.decoder_inputs().cpu().numpy().copy()
    paths = rnn_decoder.dynamic_decode.last_paths                  # (forward, backward) took the persistent launch
    got = {}

    def capture(g):
        got['denc'] = g.cpu().numpy()
        return [None]
    tape.ops[0].backward = capture
    tape.backward(loss)
    grads = {k: v.grad.cpu().numpy().copy() for k, v in store.vars.items() if v.grad is not None}
    return logits['text'].cpu().numpy(), got['denc'], grads, used, paths


@pytest.mark.parametrize('attention', ['vanilla', 'location_aware'])
def test_persistent_decoder_with_output_dropout(attention):
    """real training recipes (speller.py:36-40: DropoutWrapper(output_keep_prob)) on the persistent path: the mask
    of a step is recomputed from the Philox stream inside the persistent kernels (query and projection see the
    dropped output, the recurrence keeps h; backward: mask / keep on the output gradient only).  Same seed ->
    same masks as the step chain (NABU_SPELLER_PERSIST=0): logits and every gradient agree to the tolerance of
    the no-dropout persistent tests, and they differ from a run without dropout by O(1)."""
    import os
    from nabu_amd import ops as hip
    rng = np.random.default_rng(31)
    B, Te, E, C, U = 32, 40, 64, 8, 64
    enc_len = rng.integers(Te // 2, Te + 1, B).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(1, 9, B).astype(np.int32)
    tlen[3] = 8
    enc = rng.normal(size=(B, Te, E)).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    tg = rng.integers(0, C - 1, (B, int(tlen.max()))).astype(np.int32)
    for b in range(B):
        tg[b, tlen[b] - 1] = C - 1
        tg[b, tlen[b]:] = 0
    over = {'decoder.num_layers': 1, 'decoder.num_units': U, 'decoder.attention': attention, 'decoder.dropout': 0.7}
    if attention == 'location_aware':
        over.update({'decoder.numfilt': 3, 'decoder.filtersize': 7})
    got = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=123)
    assert got[4] == (1, 1)           # the persistent kernels really ran
    os.environ['NABU_SPELLER_PERSIST'] = os.environ['NABU_SPELLER_PERSIST_BWD'] = '0'
    try:
        ref = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=123)
        assert ref[4] == (0, 0)
    finally:
        del os.environ['NABU_SPELLER_PERSIST'], os.environ['NABU_SPELLER_PERSIST_BWD']
    hip.check_persist_status()
    plain = _decoder_run(dict(over, **{'decoder.dropout': 1.0}), enc, enc_len, tg, tlen, C, seed=123)
    assert np.abs(got[0] - plain[0]).max() > 1e-2                      # dropout really acted
    assert np.abs(got[0] - ref[0]).max() < 2e-5
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    assert rel(got[1], ref[1]) < 2e-4
    for k in ref[2]:
        assert rel(got[2][k], ref[2][k]) < 2e-4, k


@pytest.mark.parametrize('attention,dropout', [('location_aware', 0.7), ('location_aware', 1.0), ('vanilla', 0.7)])
def test_step_chain_rows16_products_with_output_dropout(attention, dropout):
    """The step chain's backward products of a 16-utterance sub-batch (rows16_kernel, gemm_skinny.hip: dq . Wq^T with the
    LSTM cell's backward pass as epilogue, dz . [Kx | Kh]^T) against the chain's older forms (NABU_SPELLER_ROWS16=0:
    gemm_skinny_fused without dropout, separate GEMM + dropout + cell kernels with it): the same masks from the same
    Philox stream, so every gradient agrees to fp32 summation order."""
    import os
    rng = np.random.default_rng(37)
    B, Te, E, C, U = 32, 40, 64, 8, 128                                # two sub-batches of 16; K = U and 4U multiples of 128
    enc_len = rng.integers(Te // 2, Te + 1, B).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(1, 9, B).astype(np.int32)
    tlen[3] = 8
    enc = rng.normal(size=(B, Te, E)).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    tg = rng.integers(0, C - 1, (B, int(tlen.max()))).astype(np.int32)
    for b in range(B):
        tg[b, tlen[b] - 1] = C - 1
        tg[b, tlen[b]:] = 0
    over = {'decoder.num_layers': 1, 'decoder.num_units': U, 'decoder.attention': attention, 'decoder.dropout': dropout}
    if attention == 'location_aware':
        over.update({'decoder.numfilt': 3, 'decoder.filtersize': 7})
    os.environ['NABU_SPELLER_PERSIST'] = os.environ['NABU_SPELLER_PERSIST_BWD'] = '0'
    try:
        got = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=77)
        os.environ['NABU_SPELLER_ROWS16'] = '0'
        ref = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=77)
    finally:
        del os.environ['NABU_SPELLER_PERSIST'], os.environ['NABU_SPELLER_PERSIST_BWD']
        os.environ.pop('NABU_SPELLER_ROWS16', None)
    assert got[4] == (0, 0) and ref[4] == (0, 0)
    assert np.abs(got[0] - ref[0]).max() < 2e-5                         # (the forward products differ the same way)
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    assert rel(got[1], ref[1]) < 2e-5
    for k in ref[2]:
        assert rel(got[2][k], ref[2][k]) < 2e-5, k


@pytest.mark.parametrize('attention,dropout', [('vanilla', 1.0), ('location_aware', 1.0), ('vanilla', 0.8)])
def test_persistent_decoder_with_scheduled_sampling(attention, dropout):
    """sample_prob > 0 (the reference default is 0.1, defaults/speller.cfg:15; 0.5 here so that many inputs are drawn)
    on the persistent path: the first slice's workgroup of every utterance evaluates the step's logits for its row and
    draws exactly what nabu_sample_ids draws for (seed, offset + t, row); the inputs travel through a 16-byte ring.
    Same seed -> the SAME decoder inputs as the step chain, logits and gradients to the tolerance of the other
    persistent tests; also together with output dropout (the projection sees the dropped output)."""
    import os
    from nabu_amd import ops as hip
    rng = np.random.default_rng(41)
    B, Te, E, C, U = 32, 33, 64, 9, 64
    enc_len = rng.integers(Te // 2, Te + 1, B).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(2, 10, B).astype(np.int32)
    tlen[3] = 9
    enc = rng.normal(size=(B, Te, E)).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    tg = rng.integers(0, C - 1, (B, int(tlen.max()))).astype(np.int32)
    for b in range(B):
        tg[b, tlen[b] - 1] = C - 1
        tg[b, tlen[b]:] = 0
    over = {'decoder.num_layers': 1, 'decoder.num_units': U, 'decoder.attention': attention, 'decoder.dropout': dropout,
            'decoder.sample_prob': 0.5}
    if attention == 'location_aware':
        over.update({'decoder.numfilt': 3, 'decoder.filtersize': 7})
    got = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=77)
    assert got[4][0] == 1                                             # the persistent forward kernel ran
    os.environ['NABU_SPELLER_PERSIST'] = os.environ['NABU_SPELLER_PERSIST_BWD'] = '0'
    try:
        ref = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=77)
        assert ref[4] == (0, 0)
    finally:
        del os.environ['NABU_SPELLER_PERSIST'], os.environ['NABU_SPELLER_PERSIST_BWD']
    hip.check_persist_status()
    teacher = np.concatenate([np.full((1, B), C - 1), tg[:, :int(tlen.max()) - 1].T], 0)
    assert 0.2 < (ref[3][1:] != teacher[1:]).mean() < 0.8              # inputs really were drawn
    np.testing.assert_array_equal(got[3], ref[3])                      # ... and the same ones on both paths
    assert np.abs(got[0] - ref[0]).max() < 2e-5
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    assert rel(got[1], ref[1]) < 2e-4
    for k in ref[2]:
        assert rel(got[2][k], ref[2][k]) < 2e-4, k


def test_persistent_decoder_regularised_at_the_cfg3_geometry():
    """the reference's training defaults (sample_prob 0.1, speller.cfg:15; output dropout 0.9) at the FULL decoder geometry
    of BASELINE configs[2] — 32 utterances, 125 encoder frames of 1024 features, 512 units, 40 classes, up to 60 steps:
    the 384-weight-register instantiation with the regularisation code compiled in — persistent launch against the step
    chain on the same seed: identical decoder inputs, logits and gradients to the persistent tests' tolerance"""
    import os
    from nabu_amd import ops as hip
    rng = np.random.default_rng(43)
    B, Te, E, C, U = 32, 125, 1024, 40, 512
    enc_len = rng.integers(Te // 2, Te + 1, B).astype(np.int32)
    enc_len[0] = Te
    tlen = rng.integers(20, 61, B).astype(np.int32)
    tlen[3] = 60
    enc = (0.3 * rng.normal(size=(B, Te, E))).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    tg = rng.integers(0, C - 1, (B, int(tlen.max()))).astype(np.int32)
    for b in range(B):
        tg[b, tlen[b] - 1] = C - 1
        tg[b, tlen[b]:] = 0
    over = {'decoder.num_layers': 1, 'decoder.num_units': U, 'decoder.attention': 'vanilla', 'decoder.dropout': 0.9,
            'decoder.sample_prob': 0.1}
    got = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=5)
    assert got[4] == (1, 1)
    os.environ['NABU_SPELLER_PERSIST'] = os.environ['NABU_SPELLER_PERSIST_BWD'] = '0'
    try:
        ref = _decoder_run(over, enc, enc_len, tg, tlen, C, seed=5)
    finally:
        del os.environ['NABU_SPELLER_PERSIST'], os.environ['NABU_SPELLER_PERSIST_BWD']
    hip.check_persist_status()
    np.testing.assert_array_equal(got[3], ref[3])
    assert np.abs(got[0] - ref[0]).max() < 5e-5
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    assert rel(got[1], ref[1]) < 3e-4
    for k in ref[2]:
        assert rel(got[2][k], ref[2][k]) < 3e-4, k


def test_persistent_decoder_status_word_is_sticky_and_reported():
    """the persistent decoder kernels' hang safety (include/nabu_hip.h, nabu_speller_fwd): ws[0] of the Speller
    workspace is their status word.  A non-zero word (what a timed-out launch leaves) makes the next launches return
    at once and ops.check_persist_status() raise; the check clears it and the following step is right again."""
    from nabu_amd import _hip, ops as hip
    rng = np.random.default_rng(11)
    enc_len = rng.integers(20, 41, 32).astype(np.int32)
    enc_len[0] = 40
    tlen = rng.integers(1, 6, 32).astype(np.int32)
    tlen[1] = 5
    ref = check_speller('vanilla', 1, 64, 0, 0, enc_len, tlen, E=64)
    hip.check_persist_status()
    buf = [v for (d, t), v in _hip.Workspace._bufs.items() if t == 'speller'][0]
    buf[:4].view(torch.int32).fill_(4 * 7 + 1)          # "block 7 gave up in the forward pass"
    try:
        check_speller('vanilla', 1, 64, 0, 0, enc_len, tlen, E=64)
        wrong = False
    except AssertionError:
        wrong = True                                     # the launches returned at once: no results
    assert wrong
    with pytest.raises(_hip.NabuHipError, match='decoder kernel timed out'):
        hip.check_persist_status()
    hip.check_persist_status()                           # cleared
    np.testing.assert_array_equal(check_speller('vanilla', 1, 64, 0, 0, enc_len, tlen, E=64), ref)


def check_speller(attention, nl, U, K, F, enc_len, tlen, E=24):
    from nabu_amd import variables as vs
    from nabu_amd.autodiff import Tape, SeqLen
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory
    from nabu_amd.neuralnetworks.trainers import loss_functions
    attention, _, prob_fn = attention.partition(':')
    prob_fn = prob_fn or 'softmax'
    rng = np.random.default_rng(U + K)
    B, C = len(enc_len), 8
    Te, Lmax = int(enc_len.max()), int(tlen.max()) + 1
    over = {'decoder.num_layers': nl, 'decoder.num_units': U, 'decoder.attention': attention,
            'decoder.probability_fn': prob_fn}
    if attention == 'location_aware':
        over.update({'decoder.numfilt': F, 'decoder.filtersize': K})
    if attention == 'windowed':
        over.update({'decoder.left_window_width': K, 'decoder.right_window_width': F})
    mc, _, _ = recipes.load_recipe('cfg3_las_vanilla', **over)
    dec = ed_decoder_factory.factory('speller')(mc, {'text': C}, None)
    enc = rng.normal(size=(B, Te, E)).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    tg = rng.integers(0, C - 1, (B, Lmax)).astype(np.int32)
    for b in range(B):
        tg[b, tlen[b] - 1] = C - 1
        tg[b, tlen[b]:] = 0
    store = vs.VariableStore(seed=3)
    dev = torch.device('cuda')
    enc_d = torch.tensor(enc, device=dev)
    from nabu_amd.autodiff import record
    src = torch.tensor(enc, device=dev)
    with vs.as_default(store), Tape() as tape:
        record([src], [enc_d], lambda g: [g])          # make 'encoded' require a gradient
        logits, lsl, _ = dec({'features': enc_d}, {'features': SeqLen(enc_len, dev)},
                             {'text': torch.tensor(tg, device=dev)}, {'text': SeqLen(tlen, dev)}, True)
        loss = loss_functions.average_cross_entropy({'text': torch.tensor(tg, device=dev)}, logits, lsl,
                                                    {'text': SeqLen(tlen, dev)})
    got = {}
    # capture d encoded through the identity op recorded above
    def capture(g):
        got['denc'] = g
        return [None]
    tape.ops[0].backward = capture
    tape.backward(loss)
    st = store.state_dict()
    p = speller_params(st, nl, attention)
    rl, rll, cache = O.speller_fwd(enc.astype(np.float64), enc_len, tg, tlen, p, attention, prob_fn,
                                   window=(K, F) if attention == 'windowed' else None)
    np.testing.assert_array_equal(rll, lsl['text'].host)
    lg = logits['text'].cpu().numpy()
    assert np.abs(lg - rl).max() < 2e-5
    for b in range(B):
        assert np.all(lg[b, tlen[b]:] == 0)                     # impute_finished: exact zeros
    rloss, dlg = O.average_cross_entropy(rl, tg, rll, tlen)
    assert abs(float(loss.item()) - rloss) / rloss < 1e-5
    rdenc, rg = O.speller_bwd(dlg, cache)
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    assert rel(got['denc'].cpu().numpy(), rdenc) < 2e-4
    for k, name in grad_names(nl, attention).items():
        g = store.vars[name].grad.cpu().numpy().astype(np.float64).reshape(rg[k].shape)
        assert rel(g, rg[k]) < 2e-4, k
    for n in range(nl):
        q = PRE + 'attention_wrapper/multi_rnn_cell/cell_%d/lstm_cell/' % n
        assert rel(store.vars[q + 'kernel'].grad.cpu().numpy(), rg['lstm'][n]['kernel']) < 2e-4, n
        assert rel(store.vars[q + 'bias'].grad.cpu().numpy(), rg['lstm'][n]['bias']) < 2e-4, n
    return lg


@pytest.mark.parametrize('recipe,over', [
    ('cfg3_las_vanilla', {}),
    ('cfg5_las_location', {'decoder.numfilt': 4, 'decoder.filtersize': 7})])
def test_las_training_trajectory_matches_oracle(recipe, over):
    from tests.test_hip_model import make_trainer, encoder_layers
    B, T = 4, 64
    # exact parity is checked with fp32 products (cfg5 ships bf16 input GEMMs: tests/test_hip_golden.py)
    over = dict(over, **{'encoder.num_units': 64, 'decoder.num_units': 32, 'trainer.batch_size': B,
                         'encoder.gemm_precision': 'f32'})
    attention = 'vanilla' if 'vanilla' in recipe else 'location_aware'
    data = SyntheticData(B, T, 40, min_frames=40, min_labels=2, max_labels=6, eos=True, time_reduction=8, seed=3234)
    tr = make_trainer(recipe, data, **over)
    losses = [float(tr.step(tr.to_device(data.batch(s))).item()) for s in range(3)]
    tr2 = make_trainer(recipe, data, **over)
    b0 = tr2.to_device(data.batch(0))
    tr2.model(b0['inputs'], b0['input_seq_length'], b0['targets'], b0['target_seq_length'], False)
    st = tr2.model.store.state_dict()
    layers = encoder_layers(st, 'Listener', 3)
    p = speller_params(st, 1, attention)

    def flat():
        v = []
        for l in layers:
            v += [l['fw_kernel'], l['fw_bias'], l['bw_kernel'], l['bw_bias']]
        keys = sorted(k for k in p if k != 'lstm')
        return v + [p[k] for k in keys] + [p['lstm'][0]['kernel'], p['lstm'][0]['bias']], keys
    vals, keys = flat()
    ms = [np.zeros_like(v) for v in vals]
    vs_ = [np.zeros_like(v) for v in vals]
    ref = []
    for s in range(3):
        b = data.batch(s)
        enc, el, caches = O.listener_fwd(b['inputs']['features'].astype(np.float64),
                                         b['input_seq_length']['features'], layers)
        lg, ll, cache = O.speller_fwd(enc, el, b['targets']['text'], b['target_seq_length']['text'], p, attention)
        loss, dlg = O.average_cross_entropy(lg, b['targets']['text'], ll, b['target_seq_length']['text'])
        ref.append(loss)
        denc, g = O.speller_bwd(dlg, cache)
        _, gl = O.listener_bwd(denc, caches)
        grads = []
        for l in gl:
            grads += [l['fw_kernel'], l['fw_bias'], l['bw_kernel'], l['bw_bias']]
        grads += [g[k] for k in keys] + [g['lstm'][0]['kernel'], g['lstm'][0]['bias']]
        vals, _ = flat()
        new = []
        for i, (v, gr) in enumerate(zip(vals, grads)):
            v2, ms[i], vs_[i] = O.clip_adam_update(v, gr, ms[i], vs_[i], s + 1, 1e-3)
            new.append(v2)
        for li, l in enumerate(layers):
            l['fw_kernel'], l['fw_bias'], l['bw_kernel'], l['bw_bias'] = new[4 * li:4 * li + 4]
        o = 4 * len(layers)
        for i, k in enumerate(keys):
            p[k] = new[o + i]
        p['lstm'][0]['kernel'], p['lstm'][0]['bias'] = new[-2], new[-1]
    rel = np.abs(np.array(losses) - np.array(ref)) / np.abs(ref)
    assert rel.max() < 1e-3, (losses, ref)
    assert rel.max() < 1e-4, (losses, ref)


def test_sample_ids_distribution_and_determinism():
    """nabu_sample_ids: Bernoulli(prob) selection, then a draw from softmax(logits)"""
    from nabu_amd import ops
    B, C = 200000, 7
    row = np.array([0.3, -1.0, 2.0, 0.0, 1.1, -3.0, 0.5], np.float32)
    lg = torch.tensor(np.tile(row, (B, 1)), device='cuda')
    teacher = torch.full((B,), 5, dtype=torch.int32, device='cuda')
    a = ops.sample_ids(lg, 1.0, 11, 3, teacher)
    b = ops.sample_ids(lg, 1.0, 11, 3, teacher)
    assert torch.equal(a, b)                                       # pure function of (seed, offset)
    assert not torch.equal(a, ops.sample_ids(lg, 1.0, 11, 4, teacher))
    freq = np.bincount(a.cpu().numpy(), minlength=C) / B
    sm = np.exp(row - row.max()); sm /= sm.sum()
    assert np.abs(freq - sm).max() < 5e-3, (freq, sm)
    assert torch.equal(ops.sample_ids(lg, 0.0, 11, 3, teacher), teacher)
    part = ops.sample_ids(lg, 0.3, 11, 3, teacher).cpu().numpy()
    changed = (part != 5).mean()                                   # P(select) * P(sample != 5)
    assert abs(changed - 0.3 * (1 - sm[5])) < 5e-3


@pytest.mark.parametrize('attention', ['vanilla', 'location_aware'])
def test_scheduled_sampling_matches_oracle_given_its_samples(attention):
    """sample_prob > 0 (speller.cfg default 0.1): the samples are random, but given the decoder
    inputs that were drawn the computation is deterministic — logits, loss and gradients must
    match the oracle run on those inputs (no gradient flows through a sample)."""
    from nabu_amd import variables as vs
    from nabu_amd.autodiff import Tape, SeqLen
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory, rnn_decoder
    from nabu_amd.neuralnetworks.trainers import loss_functions
    rng = np.random.default_rng(17)
    B, Te, E, C, U = 6, 11, 16, 8, 16
    over = {'decoder.num_layers': 1, 'decoder.num_units': U, 'decoder.attention': attention,
            'decoder.sample_prob': 0.6}
    if attention == 'location_aware':
        over.update({'decoder.numfilt': 3, 'decoder.filtersize': 5})
    mc, _, _ = recipes.load_recipe('cfg3_las_vanilla', **over)
    dec = ed_decoder_factory.factory('speller')(mc, {'text': C}, None)
    enc_len = np.array([11, 9, 11, 4, 7, 10], np.int32)
    enc = rng.normal(size=(B, Te, E)).astype(np.float32)
    enc *= (np.arange(Te)[None, :, None] < enc_len[:, None, None])
    tlen = np.array([6, 3, 5, 6, 2, 6], np.int32)
    tg = rng.integers(0, C - 1, (B, 6)).astype(np.int32)
    for b in range(B):
        tg[b, tlen[b] - 1] = C - 1
        tg[b, tlen[b]:] = 0
    store = vs.VariableStore(seed=5)
    dev = torch.device('cuda')
    tgd = torch.tensor(tg, device=dev)
    with vs.as_default(store), Tape() as tape:
        logits, lsl, _ = dec({'features': torch.tensor(enc, device=dev)}, {'features': SeqLen(enc_len, dev)},
                             {'text': tgd}, {'text': SeqLen(tlen, dev)}, True)
        loss = loss_functions.average_cross_entropy({'text': tgd}, logits, lsl, {'text': SeqLen(tlen, dev)})
    used = rnn_decoder.decoder_inputs().cpu().numpy().T            # [B,L]
    teacher = np.concatenate([np.full((B, 1), C - 1), tg[:, :5]], 1)
    assert np.all(used[:, 0] == C - 1)
    assert used.min() >= 0 and used.max() < C
    frac = (used[:, 1:] != teacher[:, 1:]).mean()
    assert 0.2 < frac < 0.9, frac                                  # ~0.6 * P(sample != target)
    tape.backward(loss)
    p = speller_params(store.state_dict(), 1, attention)
    rl, rll, cache = O.speller_fwd(enc.astype(np.float64), enc_len, tg, tlen, p, attention, dec_inputs=used)
    assert np.abs(logits['text'].cpu().numpy() - rl).max() < 2e-5
    rloss, dlg = O.average_cross_entropy(rl, tg, rll, tlen)
    assert abs(float(loss.item()) - rloss) / rloss < 1e-5
    _, rg = O.speller_bwd(dlg, cache)
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    for k, name in grad_names(1, attention).items():
        g = store.vars[name].grad.cpu().numpy().astype(np.float64).reshape(rg[k].shape)
        assert rel(g, rg[k]) < 2e-4, k
    q = PRE + 'attention_wrapper/multi_rnn_cell/cell_0/lstm_cell/'
    assert rel(store.vars[q + 'kernel'].grad.cpu().numpy(), rg['lstm'][0]['kernel']) < 2e-4
    # a second pass draws different samples (the RNG offset advances)
    with vs.as_default(store), Tape():
        dec({'features': torch.tensor(enc, device=dev)}, {'features': SeqLen(enc_len, dev)},
            {'text': tgd}, {'text': SeqLen(tlen, dev)}, True)
    assert not np.array_equal(rnn_decoder.decoder_inputs().cpu().numpy().T, used)


@pytest.mark.parametrize('C,N,W', [(7, 20000, 300), (40, 10176, 2048), (3, 255, 64), (5, 8192 + 257, 260)])
def test_scatter_rows_one_hot_gradient_is_the_ordered_sum(C, N, W):
    """dK[c] = the sum of the dz rows whose id is c, added in increasing row order (bit-exact against a float32
    sequential sum): dense hits (every chunk of 256 ids and every wave contributes, several LDS segments) — the
    compaction offsets of scatter_rows_kernel are what a stale shared counter would corrupt (one-hot input rows of
    tf's LSTMCell kernel, rnn_decoder.py:59-66)."""
    from nabu_amd import ops
    g = torch.Generator().manual_seed(C * 1000 + W)
    ids = torch.randint(0, C, (N,), generator=g, dtype=torch.int32)
    dz = torch.randn(N, W, generator=g)
    dK = torch.full((C, W), float('nan'), device='cuda')
    for _ in range(3):          # repeated launches: a latency-dependent race does not show on every run
        ops.scatter_rows(ids.cuda(), dz.cuda(), dK)
        got = dK.cpu().numpy()
        dzn, idn = dz.numpy(), ids.numpy()
        for c in range(C):
            rows = np.nonzero(idn == c)[0]
            ref = np.zeros(W, np.float32)
            for r in rows:
                ref = ref + dzn[r]
            np.testing.assert_array_equal(got[c], ref)
