"""EDDecoder base class (reference: models/ed_decoders/ed_decoder.py:9-136)."""
from abc import ABCMeta, abstractmethod

from nabu_amd import variables as vs
from nabu_amd.tools.default_conf import apply_defaults, defaults_path


class EDDecoder(object, metaclass=ABCMeta):
    '''a general decoder for an encoder decoder system: converts the high level
    features into output logits'''

    def __init__(self, conf, output_dims, constraint, name=None):
        self.conf = dict(conf.items('decoder'))
        apply_defaults(self.conf, defaults_path(__file__, self))
        self.outputs = list(output_dims.keys())
        self.output_dims = output_dims
        self.constraint = constraint
        self.scope = name or type(self).__name__

    def __call__(self, encoded, encoded_seq_length, targets, target_seq_length, is_training):
        '''Returns (logits dict, logit sequence length dict, final state)'''
        with vs.variable_scope(self.scope):
            return self._decode(encoded, encoded_seq_length, targets, target_seq_length, is_training)

    @abstractmethod
    def _decode(self, encoded, encoded_seq_length, targets, target_seq_length, is_training):
        '''create the variables and decode an entire sequence'''

    @abstractmethod
    def zero_state(self, encoded_dim, batch_size):
        '''the decoder zero state'''

    @property
    def variables(self):
        variables = vs.default_store().variables(self.scope + '/')
        if hasattr(self, 'wrapped'):
            variables += self.wrapped.variables
        return variables
