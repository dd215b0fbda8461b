"""RNN cell descriptions (reference: nabu/neuralnetworks/components/rnn_cell.py and the
tf.contrib cells the Speller stacks).  Cells here are declarative: they say which
variables exist and how a step is wired; rnn_decoder.dynamic_decode executes them
with HIP kernels."""
from nabu_amd import variables as vs


class LSTMCell(object):
    '''tf.contrib.rnn.LSTMCell: kernel [(in+U),4U] (glorot), bias zeros, gates i,j,f,o,
    forget_bias 1'''

    def __init__(self, num_units, output_keep_prob=1.0):
        self.num_units = int(num_units)
        self.output_keep_prob = float(output_keep_prob)     # DropoutWrapper(output_keep_prob)

    def variables(self, index, input_dim):
        U = self.num_units
        with vs.variable_scope('multi_rnn_cell/cell_%d/lstm_cell' % index):
            return (vs.get_variable('kernel', [input_dim + U, 4 * U]),
                    vs.get_variable('bias', [4 * U], vs.zeros))


class AttentionWrapper(object):
    '''tf.contrib.seq2seq.AttentionWrapper(cells, mechanism, output_attention=False):
    cell input = [inputs, previous context]; query = top cell output; state carries
    context ("attention") and alignments'''

    def __init__(self, cells, attention_mechanism):
        self.cells = cells
        self.attention_mechanism = attention_mechanism

    @property
    def output_size(self):
        return self.cells[-1].num_units


class AttentionProjectionWrapper(object):
    '''maps concat([cell output, context of the SAME step]) to output_dim with a Dense
    layer with bias (reference rnn_cell.py:109-155)'''

    def __init__(self, cell, output_dim, activation=None):
        if activation is not None:
            raise NotImplementedError('the Speller uses the linear projection')
        self._cell = cell
        self._output_dim = int(output_dim)

    @property
    def output_size(self):
        return self._output_dim

    def variables(self, context_dim):
        U = self._cell.output_size
        with vs.variable_scope('dense'):
            return (vs.get_variable('kernel', [U + context_dim, self._output_dim]),
                    vs.get_variable('bias', [self._output_dim], vs.zeros))
