"""Evaluator classes by recipe name (the role of nabu/neuralnetworks/evaluators/evaluator_factory.py:4-24).
decoder_evaluator (beam-search decoding + error rate) is inference, SURVEY.md 8(f) row 4."""
from nabu_amd.tools.registry import Registry

factory = Registry('evaluator', {
    'loss_evaluator': 'nabu_amd.neuralnetworks.evaluators.loss_evaluator:LossEvaluator',
}, outside=('decoder_evaluator',), undefined='Undefined %s type: %s')
