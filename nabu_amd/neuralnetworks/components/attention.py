"""Attention mechanisms (reference: nabu/neuralnetworks/components/attention.py).

The factory keeps the reference's keys ('vanilla' -> tf BahdanauAttention,
'location_aware' -> LocationAwareAttention, 'windowed' -> WindowedAttention); a mechanism object only owns its
variables — the arithmetic of a decoder step is the fused kernel nabu_attn_fwd /
nabu_attn_bwd driven by rnn_decoder.dynamic_decode."""
from nabu_amd import variables as vs

KINDS = {'vanilla': 0, 'location_aware': 1, 'windowed': 2}
PROB_FNS = {'softmax': 0, 'sigmoid': 1, 'normalized_sigmoid': 2}


def factory(conf, num_units, encoded, encoded_seq_length):
    '''create the attention mechanism (reference attention.py:6-39)'''
    if conf['probability_fn'] not in PROB_FNS:
        raise KeyError(conf['probability_fn'])              # the reference indexes a dict (attention.py:9-13)
    mech = _mechanism(conf, num_units, encoded, encoded_seq_length)
    mech.prob_fn = PROB_FNS[conf['probability_fn']]
    return mech


def _mechanism(conf, num_units, encoded, encoded_seq_length):
    if conf['attention'] == 'location_aware':
        return LocationAwareAttention(num_units=num_units, numfilt=int(conf['numfilt']),
                                      filtersize=int(conf['filtersize']), memory=encoded,
                                      memory_sequence_length=encoded_seq_length)
    elif conf['attention'] == 'vanilla':
        return BahdanauAttention(num_units=num_units, memory=encoded,
                                 memory_sequence_length=encoded_seq_length)
    elif conf['attention'] == 'windowed':
        return WindowedAttention(num_units=num_units, left_window_width=int(conf['left_window_width']),
                                 right_window_width=int(conf['right_window_width']), memory=encoded,
                                 memory_sequence_length=encoded_seq_length)
    raise Exception('unknown attention type %s' % conf['attention'])


class BahdanauAttention(object):
    '''additive attention: score = v . tanh(keys + query_layer(query)), normalize=False
    (tf.contrib.seq2seq.BahdanauAttention as used at attention.py:24-30)'''
    kind = 0
    prob_fn = 0                                  # softmax
    scope = 'bahdanau_attention'

    def __init__(self, num_units, memory, memory_sequence_length):
        self.num_units = int(num_units)
        self.memory = memory                         # [B,Te,E], rows >= length are zero
        self.memory_sequence_length = memory_sequence_length
        self.numfilt = self.filtersize = 0

    def variables(self):
        '''memory_layer / query_layer kernels (Dense, no bias) and attention_v'''
        E, U = self.memory.shape[2], self.num_units
        v = {'memory_kernel': vs.get_variable('memory_layer/kernel', [E, U])}
        with vs.variable_scope(self.scope):
            v['query_kernel'] = vs.get_variable('query_layer/kernel', [U, U])
            v['attention_v'] = vs.get_variable('attention_v', [U])
        return v


class LocationAwareAttention(BahdanauAttention):
    '''adds f = Dense(conv1d(previous alignments)) inside the tanh
    (reference attention.py:90-240)'''
    kind = 1
    scope = 'location_aware_attention'

    def __init__(self, num_units, numfilt, filtersize, memory, memory_sequence_length):
        super(LocationAwareAttention, self).__init__(num_units, memory, memory_sequence_length)
        self.numfilt, self.filtersize = int(numfilt), int(filtersize)

    def variables(self):
        v = super(LocationAwareAttention, self).variables()
        with vs.variable_scope(self.scope):
            # tf.layers.conv1d kernel [filtersize, 1, numfilt] (stored [K,F]) and the Dense on top
            v['conv_kernel'] = vs.get_variable('conv1d/kernel', [self.filtersize, 1, self.numfilt])
            v['conv_proj'] = vs.get_variable('process_conv_features/kernel', [self.numfilt, self.num_units])
        return v


class WindowedAttention(BahdanauAttention):
    '''Bahdanau attention restricted to a window around the median of the previous alignments
    (reference attention.py:294-396); initial alignments are one-hot at the first frame.  The C ABI
    carries the window widths in the descriptor fields location-aware attention uses for
    filtersize / numfilt.'''
    kind = 2
    scope = 'windowed_attention'             # tf.variable_scope(None, 'windowed_attention', ...), attention.py:374

    def __init__(self, num_units, left_window_width, right_window_width, memory, memory_sequence_length):
        super(WindowedAttention, self).__init__(num_units, memory, memory_sequence_length)
        if int(right_window_width) < 1 or int(left_window_width) < 0:
            raise Exception('windowed attention needs left_window_width >= 0 and right_window_width >= 1')
        self.filtersize, self.numfilt = int(left_window_width), int(right_window_width)
