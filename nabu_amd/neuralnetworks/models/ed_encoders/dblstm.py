"""Deep bidirectional LSTM encoder (reference: models/ed_encoders/dblstm.py:11-59)."""
from nabu_amd import variables as vs
from nabu_amd.neuralnetworks.components import layer, ops
from nabu_amd.neuralnetworks.models.ed_encoders import ed_encoder


class DBLSTM(ed_encoder.EDEncoder):
    '''A deep bidirectional LSTM classifier'''

    def encode(self, inputs, input_seq_length, is_training):
        encoded, encoded_seq_length = {}, {}
        keep = float(self.conf['dropout'])
        # build addition: arithmetic of the input-to-hidden GEMMs (BASELINE.json configs[4] asks for
        # bf16 MFMA there); 'default' = the process default = exact fp32
        layer.GEMM_PRECISION[0] = self.conf.get('gemm_precision', 'default')
        for inp in inputs:
            with vs.variable_scope(inp):
                if is_training and float(self.conf['input_noise']) > 0:      # dblstm.py:37-42
                    logits = ops.input_noise(inputs[inp], float(self.conf['input_noise']),
                                             ops.global_rng())
                else:
                    logits = inputs[inp]
                for l in range(int(self.conf['num_layers'])):                # dblstm.py:44-54
                    logits = layer.blstm(inputs=logits, sequence_length=input_seq_length[inp],
                                         num_units=int(self.conf['num_units']),
                                         scope='layer' + str(l))
                    if is_training and keep < 1:
                        logits = ops.seq_dropout(logits, keep, ops.global_rng())
                encoded[inp] = logits
                encoded_seq_length[inp] = input_seq_length[inp]
        return encoded, encoded_seq_length
