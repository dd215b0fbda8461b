"""DBLSTM: BLSTM layers at the full frame rate (the role of
nabu/neuralnetworks/models/ed_encoders/dblstm.py:11-59); one C-ABI call per layer."""
from nabu_amd import variables as vs
from nabu_amd.neuralnetworks.components import layer, ops
from nabu_amd.neuralnetworks.models.ed_encoders import ed_encoder


class DBLSTM(ed_encoder.EDEncoder):
    """cfg keys: num_layers, num_units, input_noise, dropout (keep probability), gemm_precision"""

    def encode(self, inputs, input_seq_length, is_training):
        layer.GEMM_PRECISION[0] = self.conf.get('gemm_precision', 'default')
        layer.RECURRENT_PRECISION[0] = self.conf.get('recurrent_precision', 'default')   # see listener.py
        keep, noise = float(self.conf['dropout']), float(self.conf['input_noise'])
        units = int(self.conf['num_units'])
        encoded = {}
        for name, x in inputs.items():
            with vs.variable_scope(name):
                if is_training and noise > 0:                              # dblstm.py:37-42
                    x = ops.input_noise(x, noise, ops.global_rng())
                for index in range(int(self.conf['num_layers'])):          # dblstm.py:44-54
                    x = layer.blstm(inputs=x, sequence_length=input_seq_length[name], num_units=units,
                                    scope='layer%d' % index)
                    if is_training and keep < 1:
                        x = ops.seq_dropout(x, keep, ops.global_rng())
                encoded[name] = x
        return encoded, dict(input_seq_length)
