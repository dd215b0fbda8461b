"""Trainer factory (reference: nabu/neuralnetworks/trainers/trainer_factory.py:4-17)."""


def factory(trainer):
    '''get a Trainer class by its recipe name'''
    if trainer == 'standard':
        from nabu_amd.neuralnetworks.trainers import standard_trainer
        return standard_trainer.StandardTrainer
    else:
        raise Exception('Undefined trainer type: %s' % trainer)
