"""The encoder-decoder Model (reference: nabu/neuralnetworks/models/model.py:7-86).
Same constructor, same ``__call__`` contract: dicts of batch-major tensors keyed
by io-name in, ``(logits, logit_seq_length)`` dicts out."""
from nabu_amd import variables as vs
from nabu_amd.neuralnetworks.models.ed_encoders import ed_encoder_factory
from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder_factory


class Model(object):
    '''a general class for an encoder decoder system'''

    def __init__(self, conf, trainlabels, constraint, seed=0):
        '''
        Args:
            conf: the model configuration as a ConfigParser (sections io,
                encoder, decoder — reference model.py:10-40)
            trainlabels: number of extra labels required by the trainer
            constraint: parameter constraint (None on the hot path)
            seed: seed of the parameter initialisers (not in the reference,
                where TF's graph seed plays this role)
        '''
        self.conf = conf
        self.input_names = conf.get('io', 'inputs').split(' ')
        if self.input_names == ['']:
            self.input_names = []
        self.output_names = conf.get('io', 'outputs').split(' ')
        if self.output_names == ['']:
            self.output_names = []
        # output dimensions: model.py:29-31
        self.output_dims = {}
        for i, d in enumerate(conf.get('io', 'output_dims').split(' ')):
            self.output_dims[self.output_names[i]] = int(d) + trainlabels
        self.store = vs.VariableStore(seed=seed)
        self.encoder = ed_encoder_factory.factory(conf.get('encoder', 'encoder'))(conf, constraint)
        self.decoder = ed_decoder_factory.factory(conf.get('decoder', 'decoder'))(
            conf, self.output_dims, constraint)

    def __call__(self, inputs, input_seq_length, targets, target_seq_length, is_training):
        '''Forward computation (reference model.py:42-80).

        Args: dicts keyed by io-name of [batch, time, ...] tensors / [batch] lengths.
        Returns: (logits dict, logit sequence length dict)'''
        with vs.as_default(self.store):
            encoded, encoded_seq_length = self.encoder(
                inputs=inputs, input_seq_length=input_seq_length, is_training=is_training)
            logits, logit_seq_length, _ = self.decoder(
                encoded=encoded, encoded_seq_length=encoded_seq_length, targets=targets,
                target_seq_length=target_seq_length, is_training=is_training)
        return logits, logit_seq_length

    @property
    def variables(self):
        '''the model's variables (encoder then decoder), reference model.py:82-86'''
        with vs.as_default(self.store):
            return self.encoder.variables + self.decoder.variables
