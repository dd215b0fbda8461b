"""EDEncoder base class (reference: models/ed_encoders/ed_encoder.py:9-96)."""
from abc import ABCMeta, abstractmethod

from nabu_amd import variables as vs
from nabu_amd.tools.default_conf import apply_defaults, defaults_path


class EDEncoder(object, metaclass=ABCMeta):
    '''a general encoder for an encoder decoder system: transforms input
    features into a high level representation'''

    def __init__(self, conf, constraint, name=None):
        '''conf: ConfigParser with an [encoder] section; defaults come from
        defaults/<classname>.cfg (ed_encoder.py:25-33)'''
        self.conf = dict(conf.items('encoder'))
        apply_defaults(self.conf, defaults_path(__file__, self))
        self.constraint = constraint
        self.scope = name or type(self).__name__       # tf.VariableScope name

    def __call__(self, inputs, input_seq_length, is_training):
        '''inputs: dict of [B,T,...] tensors; input_seq_length: dict of [B]
        vectors.  Returns (outputs dict, output sequence length dict).'''
        with vs.variable_scope(self.scope):
            return self.encode(inputs, input_seq_length, is_training)

    @abstractmethod
    def encode(self, inputs, input_seq_length, is_training):
        '''create the variables and do the forward computation'''

    @property
    def variables(self):
        '''variables under this encoder's scope (ed_encoder.py:84-96)'''
        variables = vs.default_store().variables(self.scope + '/')
        if hasattr(self, 'wrapped'):
            variables += self.wrapped.variables
        return variables
