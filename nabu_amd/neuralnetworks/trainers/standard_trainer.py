"""StandardTrainer (reference: nabu/neuralnetworks/trainers/standard_trainer.py:6-41)."""
from nabu_amd.neuralnetworks.trainers import trainer


class StandardTrainer(trainer.Trainer):
    '''a trainer with no added functionality'''

    def aditional_loss(self):
        '''an additional loss term, or None'''
        return None

    def chief_only_hooks(self, outputs):
        return []

    def hooks(self, outputs):
        return []
