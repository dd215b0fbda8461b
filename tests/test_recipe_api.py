"""CPU tests of the host-side recipe API (no kernel launches): cfg defaults,
factories, Model/Trainer construction, the synthetic batch contract."""
import os

import numpy as np
import pytest

from nabu_amd import recipes
from nabu_amd.tools.default_conf import apply_defaults

REF = '/root/reference'


def test_apply_defaults_fills_and_requires(tmp_path):
    f = tmp_path / 'x.cfg'
    f.write_text('[default]\na = 1\nb =\n')
    assert apply_defaults({'b': '2'}, str(f)) == {'a': '1', 'b': '2'}
    with pytest.raises(Exception, match='field b was not found'):
        apply_defaults({}, str(f))
    assert apply_defaults({'z': '0'}, str(tmp_path / 'missing.cfg')) == {'z': '0'}


def test_factories_and_unknown_names():
    from nabu_amd.neuralnetworks.models.ed_encoders import ed_encoder_factory, listener, dblstm
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory, dnn_decoder
    from nabu_amd.neuralnetworks.trainers import trainer_factory, loss_functions, standard_trainer
    assert ed_encoder_factory.factory('listener') is listener.Listener
    assert ed_encoder_factory.factory('dblstm') is dblstm.DBLSTM
    assert ed_decoder_factory.factory('dnn_decoder') is dnn_decoder.DNNDecoder
    assert trainer_factory.factory('standard') is standard_trainer.StandardTrainer
    assert loss_functions.factory('CTC') is loss_functions.CTC
    for fac, name in [(ed_encoder_factory.factory, 'nope'), (ed_decoder_factory.factory, 'nope'),
                      (trainer_factory.factory, 'nope'), (loss_functions.factory, 'nope')]:
        with pytest.raises(Exception):
            fac(name)


@pytest.mark.parametrize('recipe', ['cfg1_dblstm_ctc', 'cfg2_listener_ctc', 'cfg3_las_vanilla',
                                    'cfg5_las_location'])
def test_baseline_recipes_construct(recipe):
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    from nabu_amd.processing.synthetic import SyntheticData
    mc, tc, ec = recipes.load_recipe(recipe)
    data = SyntheticData(int(tc.get('trainer', 'batch_size')), 64, 40, batches_per_epoch=7)
    tr = trainer_factory.factory(tc.get('trainer', 'trainer'))(
        conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec, expdir=None, server=None, task_index=0)
    assert tr.model.output_dims == {'text': 40}            # 39 + trainlabels (model.py:29-31)
    assert tr.conf['numbuckets'] == '16' and tr.conf['valid_frequency'] == '500'   # defaults merged
    assert tr.train(testing=True) == []                    # graph only, like test_recipe.py of the reference
    assert tr._graph['num_steps'] == 7
    assert abs(tr.learning_rate() - 1e-3) < 1e-12


def test_unsupported_trainer_keys_raise():
    """cut_sequence_length (reference trainer.py:364-384): frame-synchronous targets only, and the reference's
    own graph construction fails on it (tf.ceil over an int32 division, trainer.py:929) -- loud, not silent"""
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    from nabu_amd.processing.synthetic import SyntheticData
    mc, tc, ec = recipes.load_recipe('cfg1_dblstm_ctc', **{'trainer.cut_sequence_length': '100'})
    with pytest.raises(Exception, match='cut_sequence_length'):
        trainer_factory.factory('standard')(conf=tc, dataconf=SyntheticData(8, 64, 40), modelconf=mc,
                                            evaluatorconf=ec, expdir=None, server=None, task_index=0)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference not mounted')
@pytest.mark.parametrize('recipe', ['DBLSTM/TIMIT', 'LAS/TIMIT', 'LAS/GP'])
def test_reference_recipe_cfgs_load_unmodified(recipe):
    """The reference's own model.cfg / trainer.cfg drive the new classes unchanged."""
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    from nabu_amd.processing.synthetic import SyntheticData
    d = os.path.join(REF, 'config', 'recipes', recipe)
    mc, tc = recipes.read_cfg(os.path.join(d, 'model.cfg')), recipes.read_cfg(os.path.join(d, 'trainer.cfg'))
    ec = recipes.from_dict({'evaluator': {'evaluator': 'None'}})
    tr = trainer_factory.factory(tc.get('trainer', 'trainer'))(
        conf=tc, dataconf=SyntheticData(4, 32, 40), modelconf=mc, evaluatorconf=ec, expdir=None,
        server=None, task_index=0)
    enc = tr.model.encoder.conf
    assert enc['num_units'] == '128' and float(enc['dropout']) == 0.5
    if recipe == 'LAS/GP':        # the windowed-attention recipe (reference attention.py:294-396)
        dec = tr.model.decoder.conf
        assert (dec['attention'], dec['left_window_width'], dec['right_window_width']) == ('windowed', '10', '15')
        assert dec['num_layers'] == '2' and tr.model.output_dims == {'phones': 48}
        assert tr.conf['loss'] == 'average_cross_entropy'
    elif recipe == 'LAS/TIMIT':
        assert tr.model.decoder.conf['attention'] == 'vanilla'       # from defaults/speller.cfg
        assert tr.model.decoder.conf['sample_prob'] == '0.1'
        assert tr.conf['loss'] == 'average_cross_entropy'
    else:
        assert tr.conf['loss'] == 'CTC' and tr.model.decoder.conf['num_layers'] == '0'


def test_synthetic_batch_contract():
    from nabu_amd.processing.synthetic import SyntheticData
    d = SyntheticData(8, 200, 40, min_frames=120, time_reduction=1, seed=1234)
    b, b2 = d.batch(3), d.batch(3)
    x, n = b['inputs']['features'], b['input_seq_length']['features']
    y, m = b['targets']['text'], b['target_seq_length']['text']
    assert x.shape == (8, 200, 40) and x.dtype == np.float32 and n.dtype == np.int32
    assert n.max() == 200 and n.min() >= 120
    for i in range(8):
        assert np.all(x[i, n[i]:] == 0) and np.any(x[i, n[i] - 1] != 0)
        lab = y[i, :m[i]]
        assert lab.max() < 39 and m[i] + np.sum(lab[1:] == lab[:-1]) <= n[i]
        assert np.all(y[i, m[i]:] == 0)
    assert np.array_equal(x, b2['inputs']['features'])                 # pure function of (seed, step)
    assert not np.array_equal(x, d.batch(4)['inputs']['features'])
    e = SyntheticData(4, 64, 40, eos=True, min_labels=3, max_labels=6).batch(0)
    for i in range(4):
        L = e['target_seq_length']['text'][i]
        assert e['targets']['text'][i, L - 1] == 39                     # EOS = C-1
    # pyramidal encoders shorten the sequence: labels must fit ceil(len/8)
    p = SyntheticData(4, 80, 40, min_frames=40, time_reduction=8, min_labels=2, max_labels=10).batch(0)
    for i in range(4):
        lab = p['targets']['text'][i, :p['target_seq_length']['text'][i]]
        assert len(lab) + np.sum(lab[1:] == lab[:-1]) <= -(-p['input_seq_length']['features'][i] // 8)


def test_decoder_and_evaluator_factories_cover_the_inference_path():
    """decoder_factory.py:4-37 / evaluator_factory.py:4-24: names of the reference dispatch to classes,
    decoders outside the hot path say so, unknown names raise 'Undefined'"""
    import configparser
    from nabu_amd.neuralnetworks.decoders import decoder_factory, decoder as decoder_mod
    from nabu_amd.neuralnetworks.evaluators import evaluator_factory
    assert decoder_factory.factory('ctc_decoder').__name__ == 'CTCDecoder'
    assert decoder_factory.factory('beam_search_decoder').__name__ == 'BeamSearchDecoder'
    assert evaluator_factory.factory('decoder_evaluator').__name__ == 'DecoderEvaluator'
    with pytest.raises(Exception, match='outside the MI355X hot path'):
        decoder_factory.factory('max_decoder')
    with pytest.raises(Exception, match='Undefined decoder type'):
        decoder_factory.factory('nope')

    class FakeModel(object):
        output_names = ['text']
        output_dims = {'text': 5}
    conf = configparser.ConfigParser()
    conf.read_dict({'decoder': {'decoder': 'beam_search_decoder', 'alphabet': 'a b c d'}})
    with pytest.raises(Exception, match='max_steps'):                  # required field (empty default)
        decoder_factory.factory('beam_search_decoder')(conf, FakeModel())
    conf.set('decoder', 'max_steps', '7')
    dec = decoder_factory.factory('beam_search_decoder')(conf, FakeModel())
    assert dec.conf['beam_width'] == '16' and dec.conf['length_penalty'] == '1' and dec.alphabet == list('abcd')
    conf2 = configparser.ConfigParser()
    conf2.read_dict({'decoder': {'decoder': 'ctc_decoder', 'text_alphabet': 'x y'}})
    ctc = decoder_factory.factory('ctc_decoder')(conf2, FakeModel())
    assert ctc.alphabets == {'text': ['x', 'y']}
    # the running error rate: (loss*num_targets + errors) / (num_targets + batch_targets)
    loss = [0.0]
    ctc._fold(loss, 3, 10)
    ctc._fold(loss, 1, 30)
    assert abs(loss[0] - 4.0 / 40.0) < 1e-12
    ctc.reset()
    assert ctc.num_targets == 0.0
    assert list(decoder_mod.host_lengths([1, 2])) == [1, 2]
