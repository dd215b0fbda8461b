"""General decoder (reference: nabu/neuralnetworks/decoders/decoder.py:8-69).

A decoder wraps a model and maps a batch of inputs to decoded sequences.  The three reference
methods are kept; where the reference's `update_evaluation_loss` returns a TF op that updates a
loss variable and a `num_targets` variable, here it folds one batch into `loss[0]` (the
1-element list of Evaluator.evaluate) and keeps the target count on the decoder — `reset()` is
the initialiser of that state."""
from abc import ABCMeta, abstractmethod

from nabu_amd.tools.default_conf import apply_defaults, defaults_path


def host_lengths(x):
    '''a [batch] length vector (SeqLen, tensor or array) as a numpy array'''
    import numpy as np
    if hasattr(x, 'host'):
        return x.host
    if hasattr(x, 'cpu'):
        return x.cpu().numpy()
    return np.asarray(x)


class Decoder(object, metaclass=ABCMeta):
    '''the abstract class for a decoder'''

    def __init__(self, conf, model):
        '''Args:
            conf: the decoder config (ConfigParser with a [decoder] section)
            model: the model that will be used for decoding'''
        self.conf = dict(conf.items('decoder'))
        apply_defaults(self.conf, defaults_path(__file__, self))
        self.model = model
        self.reset()

    def reset(self):
        '''zero the error-rate accumulator (the reference's num_targets variable)'''
        self.num_targets = 0.0

    def _fold(self, loss, errors, batch_targets):
        '''loss <- (loss*num_targets + errors) / (num_targets + batch_targets)
        (ctc_decoder.py:126-133, decoders/beam_search_decoder.py:185-194)'''
        new_num_targets = self.num_targets + float(batch_targets)
        loss[0] = (loss[0] * self.num_targets + float(errors)) / new_num_targets
        self.num_targets = new_num_targets

    @abstractmethod
    def __call__(self, inputs, input_seq_length):
        '''decode a batch: inputs / input_seq_length are io-name keyed dicts of [batch x ...]
        device tensors and [batch] lengths.  Returns the decoded sequences as a dict of outputs'''

    @abstractmethod
    def write(self, outputs, directory, names):
        '''write the outputs of the decoder for the utterances `names` under `directory`'''

    @abstractmethod
    def update_evaluation_loss(self, loss, outputs, references, reference_seq_length):
        '''fold the errors of one decoded batch into loss[0]'''
