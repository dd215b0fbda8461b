"""CPU: the validation / early-stopping / learning-rate-halving control flow of
Trainer.train (reference trainers/trainer.py:646-737, standardtrainer.cfg:32-45) with a
scripted validation loss and a stubbed training step — no GPU, no kernels."""
import numpy as np
import pytest
import torch

from nabu_amd import recipes
from nabu_amd.neuralnetworks.trainers import standard_trainer
from nabu_amd.processing.synthetic import SyntheticData


class StandardTrainer(standard_trainer.StandardTrainer):
    """the real train() loop around a fake step and a scripted validation loss (same class
    name: the defaults file is looked up by it, like in the reference)"""

    def __init__(self, script, **over):
        mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **dict({'trainer.num_epochs': 1}, **over))
        data = SyntheticData(2, 16, 40, batches_per_epoch=12)
        super().__init__(tc, data, mc, ec, None, None, 0)
        self.script = list(script)
        self.weights = 0.0                      # stands for the model + optimizer state

    def to_device(self, batch, device=None):
        return batch

    def step(self, batch):
        self.weights += 1.0
        self.last_lr = self.learning_rate()
        return torch.tensor([10.0 - 0.1 * self.global_step])

    def _ensure_variables(self):
        pass

    def validation_loss(self):
        return self.script.pop(0)

    def state(self):
        return dict(w=self.weights, global_step=self.global_step, lr=self.learning_rate_fact,
                    validated_step=self.validated_step, best=self.best_validation)

    def load_state(self, st):
        self.weights, self.global_step = st['w'], st['global_step']
        self.learning_rate_fact, self.validated_step, self.best_validation = st['lr'], st['validated_step'], st['best']


def run(script, **over):
    tr = StandardTrainer(script, **{'trainer.' + k: v for k, v in over.items()})
    hist = tr.train()
    return tr, hist


def test_validates_before_the_first_step_and_every_valid_frequency():
    tr, hist = run([5.0, 4.0, 3.0, 2.0], valid_frequency=4, num_tries='None')
    assert [s for s, _ in tr.validation_history] == [0, 4, 8]      # step 12 = num_steps: loop ended
    assert len(hist) == 12 and tr.best_validation == 3.0 and tr.num_tries == 0


def test_early_stopping_after_num_tries_restores_the_validated_model():
    # improve, then worse x3 with num_tries = 2: third worse validation terminates
    tr, hist = run([5.0, 6.0, 7.0, 8.0], valid_frequency=2, num_tries=2, reset_tries='True')
    assert [s for s, _ in tr.validation_history] == [0, 2, 4, 6]
    assert len(hist) == 6                                            # stopped at step 6 of 12
    assert tr.global_step == 0 and tr.weights == 0.0                 # restored: validated at step 0
    assert tr.best_validation == 5.0


def test_go_back_reloads_and_valid_adapt_halves_the_learning_rate():
    tr, hist = run([5.0, 6.0, 4.0, 4.5, 1.0, 0.9], valid_frequency=3, num_tries='None', go_back='True',
                   valid_adapt='True')
    # step 3: worse -> back to step 0 with lr/2 (saved again with the halved factor)
    steps = [s for s, _ in tr.validation_history]
    assert steps == [0, 3, 3, 6, 6, 9]
    lrs = [h[2] for h in hist]
    assert lrs[3] < lrs[0] * 0.75                                    # halved after the go-back
    assert hist[3][0] == 0                                           # global step went back
    assert tr.learning_rate_fact == 0.25                             # 6.0 and 4.5 were worse
    assert len(hist) == 12 + 3 + 3                                   # two go-backs of 3 steps each


def test_worse_without_go_back_keeps_training_and_counts_tries():
    tr, hist = run([5.0, 6.0, 4.0, 7.0], valid_frequency=3, num_tries=5, reset_tries='True')
    assert tr.num_tries == 1                                         # 6.0 worse (1), 4.0 better (reset 0), 7.0 worse (1)
    assert tr.best_validation == 4.0 and len(hist) == 12
    tr, _ = run([5.0, 6.0, 4.0, 7.0], valid_frequency=3, num_tries=5, reset_tries='False')
    assert tr.num_tries == 2


def test_evaluator_none_disables_validation():
    mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **{'evaluator.evaluator': 'None', 'trainer.num_epochs': 1})
    tr = standard_trainer.StandardTrainer(tc, SyntheticData(2, 16, 40, batches_per_epoch=3), mc, ec, None, None, 0)
    tr._create_graph()
    assert tr.evaluator is None


def test_loss_evaluator_config_and_factory():
    from nabu_amd.neuralnetworks.evaluators import evaluator_factory, loss_evaluator
    assert evaluator_factory.factory('loss_evaluator') is loss_evaluator.LossEvaluator
    with pytest.raises(Exception, match='Undefined evaluator'):
        evaluator_factory.factory('nope')
    mc, tc, ec = recipes.load_recipe('cfg1_dblstm_ctc')
    assert ec.get('evaluator', 'loss') == 'CTC' and ec.get('evaluator', 'evaluator') == 'loss_evaluator'
    data = SyntheticData(4, 20, 40)
    val = data.validation(3, 2)
    assert val.num_batches() == 3 and val.batch(0)['inputs']['features'].shape == (2, 20, 40)
    assert not np.array_equal(val.batch(0)['inputs']['features'], data.batch(0)['inputs']['features'][:2])
