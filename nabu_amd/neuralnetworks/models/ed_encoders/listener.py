"""Listener: a stack of pyramidal BLSTM layers topped by one plain BLSTM layer (the role of
nabu/neuralnetworks/models/ed_encoders/listener.py:14-74).  Each layer is ONE call into the C ABI
(layer.blstm -> nabu_blstm_fwd); the pyramid stacking between layers is a view of the batch-major
output buffer."""
from nabu_amd import variables as vs
from nabu_amd.neuralnetworks.components import layer, ops
from nabu_amd.neuralnetworks.models.ed_encoders import ed_encoder


class Listener(ed_encoder.EDEncoder):
    """cfg keys: num_layers (pyramidal layers), num_units, pyramid_steps, input_noise, dropout (keep
    probability), gemm_precision (build addition)"""

    def _regularise(self, x, is_training):
        keep = float(self.conf['dropout'])
        return ops.seq_dropout(x, keep, ops.global_rng()) if (is_training and keep < 1) else x

    def _encode_one(self, x, lengths, is_training):
        noise = float(self.conf['input_noise'])
        if is_training and noise > 0:                                   # listener.py:40-45
            x = ops.input_noise(x, noise, ops.global_rng())
        units, depth = int(self.conf['num_units']), int(self.conf['num_layers'])
        for index in range(depth):                                      # listener.py:49-59
            x, lengths = layer.pblstm(inputs=x, sequence_length=lengths, num_units=units,
                                      num_steps=int(self.conf['pyramid_steps']), scope='layer%d' % index)
            x = self._regularise(x, is_training)
        x = layer.blstm(inputs=x, sequence_length=lengths, num_units=units, scope='layer%d' % depth)   # :61-65
        return self._regularise(x, is_training), lengths

    def encode(self, inputs, input_seq_length, is_training):
        # arithmetic of the input-to-hidden GEMMs of the layers built below (BASELINE.json configs[4]
        # asks for bf16 MFMA there); 'default' = the process default = exact fp32
        layer.GEMM_PRECISION[0] = self.conf.get('gemm_precision', 'default')
        layer.RECURRENT_PRECISION[0] = self.conf.get('recurrent_precision', 'default')
        encoded, encoded_seq_length = {}, {}
        for name, x in inputs.items():
            with vs.variable_scope(name):
                encoded[name], encoded_seq_length[name] = self._encode_one(x, input_seq_length[name], is_training)
        return encoded, encoded_seq_length
