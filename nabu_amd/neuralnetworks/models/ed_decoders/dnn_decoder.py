"""DNN decoder (reference: models/ed_decoders/dnn_decoder.py:10-76).  With the CTC
recipe (num_layers = 0) this is the single 'outlayer' linear map applied to every
encoder frame: one MFMA GEMM with the bias fused into the epilogue."""
import torch

from nabu_amd import ops as hip
from nabu_amd import variables as vs
from nabu_amd.autodiff import record, requires_grad
from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder


def linear(inputs, num_outputs, scope):
    """tf.contrib.layers.linear: weights [F,num_outputs] xavier, biases zeros."""
    B, T, F = inputs.shape
    with vs.variable_scope(scope):
        W = vs.get_variable('weights', [F, num_outputs])
        b = vs.get_variable('biases', [num_outputs], vs.zeros)
    x = inputs if inputs.is_contiguous() else inputs.contiguous()
    x2 = x.view(B * T, F)
    out = torch.empty((B, T, num_outputs), dtype=torch.float32, device=x.device)
    hip.gemm(x2, W.data, out.view(B * T, num_outputs), bias=b.data)
    need_dx = requires_grad(inputs)

    def backward(dout):
        d2 = dout.contiguous().view(B * T, num_outputs)
        for v in (W, b):
            if v.grad is None:
                v.grad = torch.zeros_like(v.data)
        hip.gemm(x2, d2, W.grad, trans_a=True)              # dW = x^T dout
        hip.colsum(d2, b.grad)
        dx = None
        if need_dx:
            dx = torch.empty_like(x)
            hip.gemm(d2, W.data, dx.view(B * T, F), trans_b=True)   # dx = dout W^T
        return [dx]
    record([inputs], [out], backward)
    return out


class DNNDecoder(ed_decoder.EDDecoder):
    '''a DNN decoder'''

    def _decode(self, encoded, encoded_seq_length, targets, target_seq_length, is_training):
        outputs, output_seq_length = {}, {}
        first = list(encoded.keys())[0]                     # encoded.values()[0]
        for o in self.output_dims:
            with vs.variable_scope(o):
                output = encoded[first]
                if int(self.conf['num_layers']) != 0:
                    raise NotImplementedError(
                        'DNNDecoder hidden layers (relu/layer_norm) are not on the hot path; '
                        'the CTC recipe uses num_layers = 0 (DBLSTM/TIMIT/model.cfg:25)')
                output = linear(output, self.output_dims[o], 'outlayer')   # dnn_decoder.py:53-57
            outputs[o] = output
            output_seq_length[o] = encoded_seq_length[first]
        return outputs, output_seq_length, ()

    def zero_state(self, encoded_dim, batch_size):
        return ()
