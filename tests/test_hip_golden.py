"""GPU: the HIP path (through the recipe API and the C ABI) against the committed golden
fixtures of tests/golden/ — loss, every gradient of step 0, and the loss trajectory under
clip + Adam.  Nothing here executes the oracle: the expected numbers are data.  Tolerances:
per-step loss 1e-3 relative is the BASELINE.json north_star bar; fp32 kernels achieve ~1e-5,
which is what is asserted."""
import os

import numpy as np
import pytest
import torch

from nabu_amd import recipes
from nabu_amd.processing.synthetic import SyntheticData

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def unpack(fx, prefix):
    return {k[len(prefix):].replace('|', '/'): v for k, v in fx.items() if k.startswith(prefix)}


def batch_of(fx, s):
    return dict(inputs={'features': fx['x%d' % s]}, input_seq_length={'features': fx['xl%d' % s]},
                targets={'text': fx['y%d' % s]}, target_seq_length={'text': fx['yl%d' % s]})


def trainer_with_weights(recipe, over, data, weights, first_batch):
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    mc, tc, ec = recipes.load_recipe(recipe, **over)
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec,
                                             expdir=None, server=None, task_index=0)
    b0 = tr.to_device(first_batch)
    with torch.no_grad():       # creates the variables (lazily, like the reference's graph build)
        tr.model(b0['inputs'], b0['input_seq_length'], b0['targets'], b0['target_seq_length'], False)
    assert sorted(tr.model.store.state_dict()) == sorted(weights), 'variable names differ from the reference names'
    tr.model.store.load_state_dict(weights)
    return tr


def check_grads(tr, want, tol=3e-4):
    for v in tr.model.variables:
        got = v.grad.cpu().numpy().astype(np.float64).reshape(want[v.name].shape)
        ref = want[v.name].astype(np.float64)
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < tol, (v.name, err)


def step0_grads(tr, batch, loss_name):
    from nabu_amd.autodiff import Tape
    from nabu_amd.neuralnetworks.trainers import loss_functions
    b = tr.to_device(batch)
    for v in tr.model.variables:
        if v.grad is not None:
            v.grad.zero_()
    with Tape() as tape:
        logits, lsl = tr.model(b['inputs'], b['input_seq_length'], b['targets'], b['target_seq_length'], True)
        loss = getattr(loss_functions, loss_name)(b['targets'], logits, lsl, b['target_seq_length'])
    tape.backward(loss)
    loss_functions.check_status()
    return float(loss.item())


@pytest.mark.parametrize('name,recipe', [('cfg1_small', 'cfg1_dblstm_ctc'), ('cfg2_small', 'cfg2_listener_ctc')])
def test_ctc_recipes_match_golden(name, recipe):
    fx = load(name)
    B, T, D, H, nl, C, steps, seed = [int(v) for v in fx['meta']]
    over = {'encoder.num_units': H, 'encoder.num_layers': nl, 'trainer.batch_size': B}
    tr = trainer_with_weights(recipe, over, SyntheticData(B, T, D), unpack(fx, 'w:'), batch_of(fx, 0))
    loss0 = step0_grads(tr, batch_of(fx, 0), 'CTC')
    assert abs(loss0 - fx['losses'][0]) / fx['losses'][0] < 2e-5
    check_grads(tr, unpack(fx, 'g:'))
    losses = [float(tr.step(tr.to_device(batch_of(fx, s))).item()) for s in range(steps)]
    rel = np.abs(np.array(losses) - fx['losses']) / fx['losses']
    assert rel.max() < 1e-3 and rel.max() < 5e-5, (losses, fx['losses'])
    # weights after `steps` clip+Adam updates (an element whose gradient is rounding noise may
    # move by lr per step in either direction)
    wT = unpack(fx, 'wT:')
    g0 = unpack(fx, 'g:')
    st = tr.model.store.state_dict()
    for k in wT:
        d = np.abs(st[k] - wT[k])
        assert d.max() < (steps + 0.5) * 1e-3 and d.mean() < 2e-5, (k, d.max(), d.mean())
        # ... which only elements with a rounding-noise gradient may use: Adam's first steps move an element by
        # ~lr * sign(g), so wherever the oracle's step-0 gradient is clearly non-zero (> 1e-3 of the variable's
        # largest) the fp32 trajectory must stay within a small fraction of ONE step's travel of the float64 one
        sure = np.abs(g0[k].reshape(wT[k].shape)) > 1e-3 * np.abs(g0[k]).max()
        assert sure.mean() > 0.5, (k, sure.mean())                  # observed 65 - 100 % of the elements
        print('\n%s: %.0f %% of the elements have a clear gradient; their final-weight error max %.2e (all: %.2e)'
              % (k, 100 * sure.mean(), d[sure].max(), d.max()))
        assert d[sure].max() < 5e-6, (k, d[sure].max())          # observed <= 1.1e-6; one step's travel is 1e-3


@pytest.mark.parametrize('name,recipe', [('cfg3_small', 'cfg3_las_vanilla'), ('cfg5_small', 'cfg5_las_location')])
def test_las_recipes_match_golden(name, recipe):
    fx = load(name)
    B, T, D, H, nl, C, steps, seed, U, K, F = [int(v) for v in fx['meta']]
    over = {'encoder.num_units': H, 'decoder.num_units': U, 'trainer.batch_size': B,
            'encoder.gemm_precision': 'f32'}          # exact-fp32 parity; the bf16 variant is tested below
    if K:
        over.update({'decoder.numfilt': F, 'decoder.filtersize': K})
    tr = trainer_with_weights(recipe, over, SyntheticData(B, T, D, eos=True), unpack(fx, 'w:'), batch_of(fx, 0))
    loss0 = step0_grads(tr, batch_of(fx, 0), 'average_cross_entropy')
    assert abs(loss0 - fx['losses'][0]) / fx['losses'][0] < 2e-5
    check_grads(tr, unpack(fx, 'g:'))
    losses = [float(tr.step(tr.to_device(batch_of(fx, s))).item()) for s in range(steps)]
    rel = np.abs(np.array(losses) - fx['losses']) / fx['losses']
    assert rel.max() < 1e-3 and rel.max() < 1e-4, (losses, fx['losses'])


def test_cfg1_exact_matches_golden():
    """BASELINE.json configs[0] at its full size (2x256 DBLSTM, 8 x 200 x 40): 3-step loss
    trajectory, gradient norms and sampled gradient entries of step 0"""
    from tests.golden import make_golden as G       # generators only (seeded numpy); no oracle math runs
    fx = load('cfg1_exact')
    names, data = G.cfg1_exact_setup()
    w = G.draw_weights(names)
    tr = trainer_with_weights('cfg1_dblstm_ctc', {}, data, w, data.batch(0))
    loss0 = step0_grads(tr, data.batch(0), 'CTC')
    assert abs(loss0 - fx['losses'][0]) / fx['losses'][0] < 2e-5
    for v in tr.model.variables:
        g = v.grad.cpu().numpy().astype(np.float64).ravel()
        key = v.name.replace('/', '|')
        assert abs(np.sqrt((g ** 2).sum()) - fx['gnorm:' + key]) / fx['gnorm:' + key] < 2e-4, v.name
        ref = fx['gsample:' + key]
        assert np.abs(g[G.sample_index(v.name, g.size)] - ref).max() < 3e-4 * np.abs(ref).max() + 1e-6, v.name
    losses = [float(tr.step(tr.to_device(data.batch(s))).item()) for s in range(3)]
    rel = np.abs(np.array(losses) - fx['losses']) / fx['losses']
    assert rel.max() < 1e-3 and rel.max() < 5e-5, (losses, fx['losses'])


@pytest.mark.parametrize('name', ['cfg2_exact', 'cfg2_exact:f32', 'cfg2_exact:f32+rec_f32', 'cfg2_exact:bf16x6',
                                  'cfg2_exact:f16x3', 'cfg2_exact:f16x3+rec_f32', 'cfg3_exact', 'cfg5_exact'])
def test_full_size_configs_match_the_oracle_goldens(name):
    """BASELINE.json configs[1], [2] and (one GPU's share of) [4] at their FULL size — the headline
    config 32 x 1000 x 40, 4 x 512 included — against numbers the float64 oracle produced in the
    build container (tests/golden/make_golden.py <name>; cross-checked by an independent torch
    float64 implementation, tests/golden/exact_xcheck.json): loss of every step of the clip+Adam
    trajectory <= 1e-3 relative (north_star; fp32 kernels achieve ~1e-5), and for step 0 every
    variable's gradient norm and 32 sampled gradient entries <= 3e-4.  Ragged lengths (parity
    batches of SURVEY.md 8(d)).  cfg5 runs its products in exact fp32 here (the bf16 variant of the
    recipe is bounded against the fp32 goldens by the small-size test below and, at full size, by
    the 1e-3 loss bar in test_cfg5_exact_bf16_loss).
    The headline config runs under EVERY arithmetic bench.py times it in, named explicitly: 'cfg2_exact' = what the
    recipe ships (asserted below to be f16x3 products + the fp16-plane recurrence), ':f32' = exact-fp32 dense products
    (bench.py's `f32_products` leg), ':f32+rec_f32' = exact-fp32 products AND the exact-fp32 recurrent kernels (the
    `fp32_end_to_end` leg: float32 from the features to the update), ':bf16x6' (`alt_bf16x6`), ':f16x3' spelled out,
    ':f16x3+rec_f32' (the documented way out of the plane recurrence's static rounding of W_h)."""
    from tests.golden import make_golden as G       # generators only (seeded numpy); no oracle math runs
    name, _, arith = name.partition(':')             # 'cfg2_exact:bf16x6': the headline config with its dense products as
    fx = load(name)                                  # six bf16 / three scaled-fp16 plane products (gemm_pk.hip) — same tolerances
    names, data, _, loss_name, recipe, steps = G.exact_setup(name)
    w = G.draw_weights(names)
    over = {'encoder.gemm_precision': 'f32'} if name == 'cfg5_exact' else {}
    if arith:
        prec, _, rec = arith.partition('+')
        over = {'encoder.gemm_precision': prec}
        if rec:
            assert rec == 'rec_f32'
            over['encoder.recurrent_precision'] = 'f32'
    elif name == 'cfg2_exact':
        mc, _, _ = recipes.load_recipe(recipe)       # the default case IS the shipped arithmetic: say which one that is
        assert mc.get('encoder', 'gemm_precision') == 'f16x3' and not mc.has_option('encoder', 'recurrent_precision')
    tr = trainer_with_weights(recipe, over, data, w, data.batch(0))
    loss0 = step0_grads(tr, data.batch(0), loss_name)
    assert abs(loss0 - fx['losses'][0]) / fx['losses'][0] < 5e-5, (loss0, fx['losses'][0])
    for v in tr.model.variables:
        g = v.grad.cpu().numpy().astype(np.float64).ravel()
        key = v.name.replace('/', '|')
        assert abs(np.sqrt((g ** 2).sum()) - fx['gnorm:' + key]) / fx['gnorm:' + key] < 3e-4, v.name
        ref = fx['gsample:' + key]
        assert np.abs(g[G.sample_index(v.name, g.size)] - ref).max() < 3e-4 * np.abs(ref).max() + 1e-7, v.name
    losses = [float(tr.step(tr.to_device(data.batch(s))).item()) for s in range(steps)]
    rel = np.abs(np.array(losses) - fx['losses'][:steps]) / fx['losses'][:steps]
    assert rel.max() < 1e-3 and rel.max() < 1e-4, (losses, fx['losses'])


def test_cfg5_exact_bf16_loss():
    """configs[4] as shipped (bf16 input-to-hidden GEMMs) at full size 64 x 1600 x 80: the loss stays
    within the north_star tolerance of the fp32 oracle number"""
    from tests.golden import make_golden as G
    fx = load('cfg5_exact')
    names, data, _, loss_name, recipe, _ = G.exact_setup('cfg5_exact')
    tr = trainer_with_weights(recipe, {}, data, G.draw_weights(names), data.batch(0))
    loss0 = step0_grads(tr, data.batch(0), loss_name)
    assert abs(loss0 - fx['losses'][0]) / fx['losses'][0] < 1e-3, (loss0, fx['losses'][0])


def test_cfg5_recipe_with_its_bf16_input_gemms_stays_within_the_north_star_tolerance():
    """BASELINE.json configs[4]: the cfg5 recipe as shipped (encoder.gemm_precision = bf16: operands of
    the input-to-hidden products rounded to bf16, fp32 accumulation and state) against the fp32
    golden trajectory: per-step loss within 1e-3 relative (north_star), gradients within 3 %."""
    fx = load('cfg5_small')
    B, T, D, H, nl, C, steps, seed, U, K, F = [int(v) for v in fx['meta']]
    over = {'encoder.num_units': H, 'decoder.num_units': U, 'trainer.batch_size': B,
            'decoder.numfilt': F, 'decoder.filtersize': K}
    mc, _, _ = recipes.load_recipe('cfg5_las_location')
    assert mc.get('encoder', 'gemm_precision') == 'bf16'
    tr = trainer_with_weights('cfg5_las_location', over, SyntheticData(B, T, D, eos=True), unpack(fx, 'w:'),
                              batch_of(fx, 0))
    loss0 = step0_grads(tr, batch_of(fx, 0), 'average_cross_entropy')
    assert abs(loss0 - fx['losses'][0]) / fx['losses'][0] < 1e-3
    assert abs(loss0 - fx['losses'][0]) > 0          # really a different arithmetic
    check_grads(tr, unpack(fx, 'g:'), tol=3e-2)
    losses = [float(tr.step(tr.to_device(batch_of(fx, s))).item()) for s in range(steps)]
    rel = np.abs(np.array(losses) - fx['losses']) / fx['losses']
    assert rel.max() < 1e-3, (losses, fx['losses'])


def test_decoders_against_the_golden_decode_fixture():
    """inference kernels (decode.hip) against committed vectors: CTC beam search with and without
    merge_repeated, edit distance, attention beam search (vanilla and location-aware)"""
    from nabu_amd import ops
    from nabu_amd import variables as vs
    from nabu_amd.autodiff import SeqLen
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory, rnn_decoder
    fx = load('decode')
    dev = 'cuda'
    lg, ln = torch.tensor(fx['ctc_logits'], device=dev), torch.tensor(fx['ctc_lens'], device=dev)
    for merge in (1, 0):
        ids, out_len, _ = ops.ctc_beam_search(lg, ln, 100, bool(merge))
        np.testing.assert_array_equal(out_len.cpu().numpy(), fx['ctc_len_merge%d' % merge])
        np.testing.assert_array_equal(ids.cpu().numpy(), fx['ctc_ids_merge%d' % merge])
    ids, out_len, _ = ops.ctc_beam_search(lg, ln, 100, True)
    dist = ops.edit_distance(ids, out_len, torch.tensor(fx['ed_ref'], device=dev),
                             torch.tensor(fx['ed_ref_len'], device=dev))
    np.testing.assert_array_equal(dist.cpu().numpy(), fx['ed_dist'])
    for attention, K, F in (('vanilla', 0, 0), ('location_aware', 5, 3)):
        pre = 'bs_%s_' % attention
        w = unpack(fx, pre + 'w_')
        over = {'decoder.num_layers': 1, 'decoder.num_units': 32, 'decoder.attention': attention}
        if K:
            over.update({'decoder.numfilt': F, 'decoder.filtersize': K})
        mc, _, _ = recipes.load_recipe('cfg3_las_vanilla', **over)
        dec = ed_decoder_factory.factory('speller')(mc, {'text': 9}, None)
        store = vs.VariableStore(seed=0)
        store.restore_from(w)
        enc, enc_len = torch.tensor(fx[pre + 'enc'], device=dev), SeqLen(fx[pre + 'enc_len'], dev)
        with torch.no_grad(), vs.as_default(store), vs.variable_scope(dec.scope):
            cell = dec.create_cell({'features': enc}, {'features': enc_len}, False)
            seqs, lengths, scores, aligns = rnn_decoder.beam_search(cell, enc, enc_len, 6, 12, 1.0, 1.0)
        assert not store.restore, 'variable names differ from the reference names: %s' % sorted(store.restore)
        np.testing.assert_array_equal(seqs.cpu().numpy(), fx[pre + 'seq'])
        np.testing.assert_array_equal(lengths.cpu().numpy(), fx[pre + 'len'])
        np.testing.assert_allclose(scores.cpu().numpy(), fx[pre + 'scores'], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(aligns.cpu().numpy(), fx[pre + 'align'], atol=2e-5)
