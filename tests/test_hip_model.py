"""GPU parity of the whole training step through the recipe API (Model / Trainer /
factories driven by reference-format cfgs) against the float64 oracle: loss,
every gradient, and the loss trajectory over several clip+Adam updates on the
same seeded synthetic batches.  Tolerance for the per-step loss: 1e-3 relative
(BASELINE.json north_star); observed errors are ~1e-6."""
import numpy as np
import pytest
import torch

from oracle import nabu_oracle as O
from nabu_amd import recipes
from nabu_amd.processing.synthetic import SyntheticData

pytestmark = pytest.mark.gpu

CELL = 'bidirectional_rnn/%s/layer_norm_basic_lstm_cell/%s'


def make_trainer(recipe, data, **over):
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    mc, tc, ec = recipes.load_recipe(recipe, **over)
    return trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec,
                                               expdir=None, server=None, task_index=0)


def encoder_layers(state, enc, num_layers):
    """TF-named variables -> oracle layer dicts (float64)."""
    layers = []
    n = num_layers + 1 if enc == 'Listener' else num_layers
    for l in range(n):
        pyr = enc == 'Listener' and l < num_layers
        pre = '%s/features/layer%d/%s' % (enc, l, 'BLSTM/' if pyr else '')
        layers.append({'%s_%s' % (d, w): state[pre + CELL % (d, w)].astype(np.float64)
                       for d in ('fw', 'bw') for w in ('kernel', 'bias')})
    return layers


def oracle_ctc_step(batch, layers, W, b, enc):
    x = batch['inputs']['features'].astype(np.float64)
    lens = batch['input_seq_length']['features']
    if enc == 'Listener':
        e, el, caches = O.listener_fwd(x, lens, layers)
    else:
        e, el, caches = O.dblstm_fwd(x, lens, layers)
    lg = O.linear_fwd(e, W, b)
    y, yl = batch['targets']['text'], batch['target_seq_length']['text']
    nll, dlg = O.ctc_loss(lg, el, y, yl)
    B = x.shape[0]
    de, dW, db = O.linear_bwd(dlg / B, e, W)
    if enc == 'Listener':
        _, grads = O.listener_bwd(de, caches)
    else:
        _, grads = O.dblstm_bwd(de, caches)
    return nll.mean(), grads, dW, db


def flat_oracle(layers, W, b):
    vals = []
    for l in layers:
        vals += [l['fw_kernel'], l['fw_bias'], l['bw_kernel'], l['bw_bias']]
    return vals + [W, b]


@pytest.mark.parametrize('recipe,enc,over,T,minT,red', [
    ('cfg1_dblstm_ctc', 'DBLSTM', {'encoder.num_units': 32, 'trainer.batch_size': 4}, 40, 25, 1),
    ('cfg2_listener_ctc', 'Listener', {'encoder.num_units': 32, 'trainer.batch_size': 4}, 64, 40, 8),
    ('cfg2_listener_ctc', 'Listener', {'encoder.num_units': 16, 'encoder.num_layers': 2,
                                       'trainer.batch_size': 5}, 45, 30, 4),      # odd T: padded pyramid
])
def test_training_trajectory_matches_oracle(recipe, enc, over, T, minT, red):
    B = over['trainer.batch_size']
    data = SyntheticData(B, T, 40, min_frames=minT, min_labels=2, max_labels=5, time_reduction=red, seed=2234)
    tr = make_trainer(recipe, data, **over)
    nl = int(tr.model.encoder.conf['num_layers'])
    losses, ref_losses = [], []
    for step in range(4):
        batch = data.batch(step)
        loss = tr.step(tr.to_device(batch))
        losses.append(float(loss.item()))
    # replay on the oracle from the same initial weights: re-create the model with the same seed
    tr2 = make_trainer(recipe, data, **over)
    b0 = tr2.to_device(data.batch(0))
    with torch.no_grad():
        tr2.model(b0['inputs'], b0['input_seq_length'], b0['targets'], b0['target_seq_length'], False)
    st = tr2.model.store.state_dict()
    layers = encoder_layers(st, enc, nl)
    W = st['DNNDecoder/text/outlayer/weights'].astype(np.float64)
    bb = st['DNNDecoder/text/outlayer/biases'].astype(np.float64)
    ms = [np.zeros_like(v) for v in flat_oracle(layers, W, bb)]
    vs_ = [np.zeros_like(v) for v in flat_oracle(layers, W, bb)]
    for step in range(4):
        loss, grads, dW, db = oracle_ctc_step(data.batch(step), layers, W, bb, enc)
        ref_losses.append(loss)
        params = flat_oracle(layers, W, bb)
        gl = flat_oracle(grads, dW, db)
        new = []
        for i, (p, g) in enumerate(zip(params, gl)):
            p2, ms[i], vs_[i] = O.clip_adam_update(p, g, ms[i], vs_[i], step + 1, 1e-3)
            new.append(p2)
        for li, l in enumerate(layers):
            l['fw_kernel'], l['fw_bias'], l['bw_kernel'], l['bw_bias'] = new[4 * li:4 * li + 4]
        W, bb = new[-2], new[-1]
    rel = np.abs(np.array(losses) - np.array(ref_losses)) / np.abs(ref_losses)
    assert rel.max() < 1e-3, (losses, ref_losses)
    assert rel.max() < 5e-5, (losses, ref_losses)           # what fp32 kernels actually achieve
    # final weights after 4 updates
    st = tr.model.store.state_dict()
    got = encoder_layers(st, enc, nl)
    for a, b_ in zip(got, layers):
        for k in a:
            # Adam moves a weight by ~lr per step whatever the gradient's size, so an
            # element whose gradient is rounding noise may differ by up to steps*lr: the MAX bound
            # (4.5e-3 = the full distance 4 steps can travel) only catches gross errors — it is the
            # MEAN bound (2e-5, i.e. almost every element agrees to float32 rounding) and the per-step
            # loss bound above that carry this check
            d = np.abs(a[k] - b_[k])
            assert d.max() < 4.5e-3 and d.mean() < 2e-5, (k, d.max(), d.mean())


def test_single_step_gradients_match_oracle():
    B, T = 4, 48
    data = SyntheticData(B, T, 40, min_frames=30, min_labels=2, max_labels=5, time_reduction=4, seed=7)
    over = {'encoder.num_units': 16, 'encoder.num_layers': 2, 'trainer.batch_size': B}
    tr = make_trainer('cfg2_listener_ctc', data, **over)
    from nabu_amd.autodiff import Tape
    from nabu_amd.neuralnetworks.trainers import loss_functions
    b = tr.to_device(data.batch(0))
    with Tape() as tape:
        logits, lsl = tr.model(b['inputs'], b['input_seq_length'], b['targets'], b['target_seq_length'], True)
        loss = loss_functions.CTC(b['targets'], logits, lsl, b['target_seq_length'])
    assert list(lsl['text'].host) == list(-(-data.batch(0)['input_seq_length']['features'] // 4))
    tape.backward(loss)
    loss_functions.check_status()
    st = tr.model.store.state_dict()
    layers = encoder_layers(st, 'Listener', 2)
    W = st['DNNDecoder/text/outlayer/weights'].astype(np.float64)
    bb = st['DNNDecoder/text/outlayer/biases'].astype(np.float64)
    rl, grads, dW, db = oracle_ctc_step(data.batch(0), layers, W, bb, 'Listener')
    assert abs(float(loss.item()) - rl) / rl < 1e-5
    names = [v.name for v in tr.model.variables]
    assert names[0] == 'Listener/features/layer0/BLSTM/' + CELL % ('fw', 'kernel')
    assert 'Listener/features/layer2/' + CELL % ('bw', 'bias') in names
    gv = {v.name: v.grad.cpu().numpy().astype(np.float64) for v in tr.model.variables}
    for l, g in enumerate(grads):
        pre = 'Listener/features/layer%d/%s' % (l, 'BLSTM/' if l < 2 else '')
        for d in ('fw', 'bw'):
            for w in ('kernel', 'bias'):
                ref = g['%s_%s' % (d, w)]
                got = gv[pre + CELL % (d, w)]
                assert np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12) < 2e-4, (l, d, w)
    assert np.abs(gv['DNNDecoder/text/outlayer/weights'] - dW).max() / np.abs(dW).max() < 2e-4
    assert np.abs(gv['DNNDecoder/text/outlayer/biases'] - db).max() / np.abs(db).max() < 2e-4


def test_repeated_updates_on_one_batch_reduce_the_loss():
    data = SyntheticData(4, 64, 40, min_frames=40, min_labels=2, max_labels=5, time_reduction=8, seed=3)
    tr = make_trainer('cfg2_listener_ctc', data, **{'encoder.num_units': 32, 'trainer.batch_size': 4})
    b = tr.to_device(data.batch(0))
    ls = [float(tr.step(b).item()) for _ in range(8)]
    assert ls[-1] < ls[0] and all(np.isfinite(ls))


def test_ctc_infeasible_batch_raises_like_tf():
    from nabu_amd.neuralnetworks.trainers import loss_functions
    data = SyntheticData(2, 16, 40, min_labels=2, max_labels=3, time_reduction=1, seed=1)
    tr = make_trainer('cfg2_listener_ctc', data, **{'encoder.num_units': 16, 'trainer.batch_size': 2})
    batch = data.batch(0)
    batch['targets']['text'] = np.zeros((2, 3), np.int32)          # 3 repeats need 5 frames, encoder has 2
    batch['target_seq_length']['text'] = np.array([3, 3], np.int32)
    tr.step(tr.to_device(batch))
    with pytest.raises(Exception, match='Not enough time'):
        loss_functions.check_status()


def test_dropout_and_input_noise_paths_run_and_are_regenerable():
    from nabu_amd import ops
    x = torch.randn(1000, device='cuda')
    y1, y2 = ops.dropout(x, 0.5, 3, 9), ops.dropout(x, 0.5, 3, 9)
    assert torch.equal(y1, y2)
    kept = (y1 != 0).float().mean().item()
    assert 0.4 < kept < 0.6 and torch.allclose(y1[y1 != 0], 2 * x[y1 != 0])
    z = ops.gaussian_noise(torch.zeros(200000, device='cuda'), 0.6, 1, 1)
    assert abs(z.mean().item()) < 0.01 and abs(z.std().item() - 0.6) < 0.01
    data = SyntheticData(3, 32, 40, min_labels=2, max_labels=3, time_reduction=8, seed=5)
    tr = make_trainer('cfg2_listener_ctc', data, **{'encoder.num_units': 16, 'encoder.input_noise': 0.6,
                                                    'encoder.dropout': 0.5, 'trainer.batch_size': 3})
    l0 = float(tr.step(tr.to_device(data.batch(0))).item())
    assert np.isfinite(l0)
    # noise on raw features depends on no parameter: its output must not ask the first layer for an input gradient
    # (two [B T, 8H] x [8H, D] products per step); on a tensor that does depend on one it is recorded
    from nabu_amd.autodiff import Tape, record, requires_grad
    from nabu_amd.neuralnetworks.components import ops as cops
    with Tape():
        raw = torch.randn(2, 8, 40, device='cuda')
        assert not requires_grad(cops.input_noise(raw, 0.6, cops.RngState(1)))
        dep = raw.clone()
        record([raw], [dep], lambda g: [None])
        assert requires_grad(cops.input_noise(dep, 0.6, cops.RngState(1)))


def test_dnn_decoder_hidden_layers_match_oracle():
    """dnn_decoder.py:40-51 with its shipped defaults (num_layers=1, layer_norm=True): names,
    logits and every gradient against the oracle"""
    from nabu_amd import variables as vs
    from nabu_amd.autodiff import Tape, SeqLen, record
    from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory
    rng = np.random.default_rng(8)
    B, T, E, C, Hd = 3, 7, 12, 6, 8
    mc, _, _ = recipes.load_recipe('cfg2_listener_ctc', **{'decoder.num_layers': 2, 'decoder.num_units': Hd,
                                                           'decoder.layer_norm': 'True'})
    dec = ed_decoder_factory.factory('dnn_decoder')(mc, {'text': C}, None)
    enc = rng.normal(size=(B, T, E)).astype(np.float32)
    dlog = rng.normal(size=(B, T, C)).astype(np.float32)
    store = vs.VariableStore(seed=2)
    dev = torch.device('cuda')
    src = torch.tensor(enc, device=dev)
    got = {}
    with vs.as_default(store), Tape() as tape:
        e = src.clone()
        record([src], [e], lambda g: [got.setdefault('denc', g)] and [None])
        logits, lsl, _ = dec({'features': e}, {'features': SeqLen(np.full(B, T, np.int32), dev)}, None, None, True)
        # perturb gamma/beta away from their (1, 0) initial values so that their use is tested
        for n in ('DNNDecoder/text/LayerNorm/gamma', 'DNNDecoder/text/LayerNorm_1/beta'):
            store.vars[n].data += torch.tensor(rng.normal(size=Hd).astype(np.float32), device=dev) * 0.3
    with vs.as_default(store), Tape() as tape:
        e = src.clone()
        record([src], [e], lambda g: [got.setdefault('denc', g)] and [None])
        logits, lsl, _ = dec({'features': e}, {'features': SeqLen(np.full(B, T, np.int32), dev)}, None, None, True)
        loss = (logits['text'] * torch.tensor(dlog, device=dev)).sum()
        record([logits['text']], [loss], lambda g: [torch.tensor(dlog, device=dev)])
    tape.backward(loss)
    st = {k: v.astype(np.float64) for k, v in store.state_dict().items()}
    assert sorted(st) == sorted(['DNNDecoder/text/%s' % n for n in (
        'layer0/weights', 'layer0/biases', 'LayerNorm/beta', 'LayerNorm/gamma', 'layer1/weights', 'layer1/biases',
        'LayerNorm_1/beta', 'LayerNorm_1/gamma', 'outlayer/weights', 'outlayer/biases')])
    P = 'DNNDecoder/text/'
    x = enc.astype(np.float64)
    caches = []
    for l, ln in ((0, 'LayerNorm'), (1, 'LayerNorm_1')):
        a = O.linear_fwd(x, st[P + 'layer%d/weights' % l], st[P + 'layer%d/biases' % l])
        r = O.relu_fwd(a)
        y, c = O.layer_norm_fwd(r, st[P + ln + '/gamma'], st[P + ln + '/beta'])
        caches.append((x, r, c))
        x = y
    ref = O.linear_fwd(x, st[P + 'outlayer/weights'], st[P + 'outlayer/biases'])
    assert np.abs(logits['text'].cpu().numpy() - ref).max() < 2e-5
    rel = lambda a, b_: np.abs(a - b_).max() / (np.abs(b_).max() + 1e-12)
    g = lambda n: store.vars[P + n].grad.cpu().numpy().astype(np.float64)
    d, dW, db = O.linear_bwd(dlog.astype(np.float64), x, st[P + 'outlayer/weights'])
    assert rel(g('outlayer/weights'), dW) < 2e-4 and rel(g('outlayer/biases'), db) < 2e-4
    for l, ln in ((1, 'LayerNorm_1'), (0, 'LayerNorm')):
        xin, r, c = caches[l]
        d, dg, dbeta = O.layer_norm_bwd(d, c)
        assert rel(g(ln + '/gamma'), dg) < 2e-4 and rel(g(ln + '/beta'), dbeta) < 2e-4
        d = O.relu_bwd(d, r)
        d, dW, db = O.linear_bwd(d, xin, st[P + 'layer%d/weights' % l])
        assert rel(g('layer%d/weights' % l), dW) < 2e-4 and rel(g('layer%d/biases' % l), db) < 2e-4
    assert rel(got['denc'].cpu().numpy(), d) < 2e-4


def test_train_loop_validates_checkpoints_and_resumes(tmp_path):
    """Trainer.train with the LossEvaluator (reference loss_evaluator.py:8-64) on the HIP forward
    path, the validated-model hook, the final model files, and resume from logdir/model.ckpt:
    an interrupted run continued from its checkpoint reproduces the uninterrupted run bit for bit"""
    import os
    over = {'encoder.num_units': 16, 'trainer.batch_size': 3, 'trainer.num_epochs': 1,
            'trainer.valid_frequency': 3, 'evaluator.batch_size': 2, 'evaluator.numbatches': 2}

    def trainer(expdir, nb):
        data = SyntheticData(3, 32, 40, min_frames=20, min_labels=2, max_labels=3, time_reduction=8, seed=11,
                             batches_per_epoch=nb)
        from nabu_amd.neuralnetworks.trainers import trainer_factory
        mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **over)
        return trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec,
                                                   expdir=expdir, server=None, task_index=0)
    full = trainer(str(tmp_path / 'full'), 6)
    hist = full.train()
    assert [h[0] for h in hist] == list(range(6))
    assert [s for s, _ in full.validation_history] == [0, 3]
    v0 = full.validation_history[0][1]
    assert np.isfinite(v0) and v0 > 0
    for f in ('model/network.ckpt.npz', 'model/model.pkl', 'logdir/validated.ckpt', 'logdir/model.ckpt'):
        assert os.path.exists(str(tmp_path / 'full' / f)), f
    # the validation loss is the utterance-weighted mean of the per-batch losses: recompute it
    ev = full.evaluator
    loss, update, nb = ev.evaluate()
    for i in range(nb):
        update(i)
    assert nb == 2 and np.isfinite(loss[0])
    # interrupted run: 3 steps (num_steps must be the same for the learning-rate schedule), resume
    part = trainer(str(tmp_path / 'part'), 6)
    part._create_graph()
    part._graph['num_steps'] = 6
    part.checkpoint_steps = 3
    orig = type(part).step
    calls = {'n': 0}

    class Stop(Exception):
        pass

    def step_then_stop(self, batch):
        if calls['n'] == 3:
            raise Stop()
        calls['n'] += 1
        return orig(self, batch)
    type(part).step = step_then_stop
    try:
        with pytest.raises(Stop):
            part.train()
    finally:
        type(part).step = orig
    assert os.path.exists(str(tmp_path / 'part' / 'logdir' / 'model.ckpt'))
    from nabu_amd.neuralnetworks.components import ops as nops
    cont = trainer(str(tmp_path / 'part'), 6)
    hist2 = cont.train()
    assert [h[0] for h in hist2] == [3, 4, 5]
    np.testing.assert_array_equal(np.array([h[1] for h in hist2]), np.array([h[1] for h in hist[3:]]))
    a, b = full.model.store.state_dict(), cont.model.store.state_dict()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])


def test_training_from_tfrecord_data_sets(tmp_path):
    """the trainer reads the reference's on-disk format (TFRecord files written by
    Array/StringWriter + metadata) through database.conf sections, with length bucketing, and the
    LossEvaluator reads the dev sections"""
    from tests.test_data_path import make_dataset
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    conf, feats, texts, alphabet = make_dataset(str(tmp_path / 'train'), n=24, dim=40, min_frames=14)
    dev, _, _, _ = make_dataset(str(tmp_path / 'dev'), n=6, dim=40, seed=5, min_frames=14)
    conf.read_dict({'devfbank': dict(dev.items('trainfbank')), 'devtext': dict(dev.items('traintext'))})
    mc, tc, ec = recipes.load_recipe('cfg1_dblstm_ctc', **{
        'encoder.num_units': 16, 'trainer.batch_size': 4, 'trainer.numbuckets': 2, 'trainer.num_epochs': 1,
        'trainer.valid_frequency': 3, 'evaluator.batch_size': 2, 'io.output_dims': 4})
    tc.set('trainer', 'features', 'trainfbank')
    tc.set('trainer', 'targets', 'text')
    tc.set('trainer', 'text', 'traintext')
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=conf, modelconf=mc, evaluatorconf=ec,
                                             expdir=None, server=None, task_index=0)
    hist = tr.train()
    assert len(hist) == tr.data.num_batches() and all(np.isfinite(h[1]) for h in hist)
    assert [s for s, _ in tr.validation_history][:2] == [0, 3]
    assert all(np.isfinite(v) and v > 0 for _, v in tr.validation_history)


def test_stale_value_bound_is_caught_in_debug_mode():
    """a tensor that carries an a-priori bound (|LSTM output| <= 1) and is then changed IN PLACE would make the next layer's
    f16x3 packs overflow silently (the bound is a promise, components/ops.py): NABU_CHECK_X_BOUND=1 measures max|x| at every
    layer call and raises"""
    from nabu_amd.neuralnetworks.components import layer, ops
    from nabu_amd import variables as vs
    store = vs.VariableStore(seed=3)
    x = torch.randn(4, 32, 40, device='cuda')
    lens = np.full(4, 32, np.int32)
    layer.CHECK_X_BOUND[0] = True
    try:
        with vs.as_default(store):
            y = layer.blstm(x, lens, 128, scope='a')
            assert ops.value_bound(y) == 1.0
            z = layer.blstm(y, lens, 128, scope='b')          # within the bound: fine
            y.mul_(3.0)                                        # the promise is broken
            with pytest.raises(RuntimeError, match='exceeds its recorded bound'):
                layer.blstm(y, lens, 128, scope='b')
        assert torch.isfinite(z).all()
    finally:
        layer.CHECK_X_BOUND[0] = False
