"""Trainer classes by recipe name (the role of nabu/neuralnetworks/trainers/trainer_factory.py:4-17)."""
from nabu_amd.tools.registry import Registry

factory = Registry('trainer', {
    'standard': 'nabu_amd.neuralnetworks.trainers.standard_trainer:StandardTrainer',
}, undefined='Undefined %s type: %s')
