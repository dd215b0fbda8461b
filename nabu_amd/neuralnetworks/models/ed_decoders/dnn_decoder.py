"""DNN decoder (reference: models/ed_decoders/dnn_decoder.py:10-76).  With the CTC
recipe (num_layers = 0) this is the single 'outlayer' linear map applied to every
encoder frame: one MFMA GEMM with the bias fused into the epilogue.  Hidden layers
(fully_connected = linear + ReLU, optional layer_norm, dropout; :40-51) run on the same
GEMM plus the relu / layer_norm kernels of the C ABI."""
import torch

from nabu_amd import ops as hip
from nabu_amd import variables as vs
from nabu_amd.autodiff import record, requires_grad
from nabu_amd.neuralnetworks.components import ops as nops
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
    record([inputs], [out], backward, params=(W, b))
    return out


def ones(rng, shape):
    import numpy as np
    return np.ones(shape, np.float32)


def relu(x):
    y = hip.relu(x if x.is_contiguous() else x.contiguous())
    record([x], [y], lambda dy: [hip.relu_bwd(y, dy.contiguous())])
    return y


def layer_norm(inputs, scope='LayerNorm'):
    """tf.contrib.layers.layer_norm(inputs) with its TF-1.8 defaults (begin_norm_axis=1,
    begin_params_axis=-1): variables <scope>/beta (zeros) and <scope>/gamma (ones), both [F];
    the moments of a [B,T,F] input are taken over (T,F) of each batch row."""
    F = inputs.shape[-1]
    with vs.variable_scope(scope):
        beta = vs.get_variable('beta', [F], vs.zeros)
        gamma = vs.get_variable('gamma', [F], ones)
    x = inputs if inputs.is_contiguous() else inputs.contiguous()
    y, mean, rstd = hip.layer_norm_fwd(x, gamma.data, beta.data)

    def backward(dy):
        dx, dgp, dbp = hip.layer_norm_bwd(x, gamma.data, dy.contiguous(), mean, rstd)
        for v in (beta, gamma):
            if v.grad is None:
                v.grad = torch.zeros_like(v.data)
        hip.colsum(dgp, gamma.grad)
        hip.colsum(dbp, beta.grad)
        return [dx]
    record([inputs], [y], backward)
    return y


class DNNDecoder(ed_decoder.EDDecoder):
    '''a DNN decoder'''

    def _decode(self, encoded, encoded_seq_length, targets, target_seq_length, is_training):
        outputs, output_seq_length = {}, {}
        first = list(encoded.keys())[0]                     # encoded.values()[0]
        for o in self.output_dims:
            with vs.variable_scope(o):
                output = encoded[first]
                for l in range(int(self.conf['num_layers'])):           # dnn_decoder.py:40-51
                    # tf.contrib.layers.fully_connected: linear + ReLU
                    output = relu(linear(output, int(self.conf['num_units']), 'layer%d' % l))
                    if self.conf['layer_norm'] == 'True':
                        # TF uniquifies the default scope name of repeated calls
                        output = layer_norm(output, 'LayerNorm' if l == 0 else 'LayerNorm_%d' % l)
                    if float(self.conf['dropout']) < 1 and is_training:
                        output = nops.seq_dropout(output, float(self.conf['dropout']), nops.global_rng())
                output = linear(output, self.output_dims[o], 'outlayer')   # dnn_decoder.py:53-57
            outputs[o] = output
            output_seq_length[o] = encoded_seq_length[first]
        return outputs, output_seq_length, ()

    def zero_state(self, encoded_dim, batch_size):
        return ()
