"""Evaluator classes by recipe name (the role of nabu/neuralnetworks/evaluators/evaluator_factory.py:4-24)."""
from nabu_amd.tools.registry import Registry

factory = Registry('evaluator', {
    'loss_evaluator': 'nabu_amd.neuralnetworks.evaluators.loss_evaluator:LossEvaluator',
    'decoder_evaluator': 'nabu_amd.neuralnetworks.evaluators.decoder_evaluator:DecoderEvaluator',
}, undefined='Undefined %s type: %s')
