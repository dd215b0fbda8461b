"""General recurrent decoder (reference: models/ed_decoders/rnn_decoder.py:13-127).

``_decode`` keeps the reference's structure — pad the targets with the start
label C-1, build the cell, run it over the whole target sequence — but
``tf.contrib.seq2seq.dynamic_decode`` over a ScheduledEmbeddingTrainingHelper is
``dynamic_decode`` below: ONE call into the C ABI (nabu_speller_fwd) whose C++
driver launches, per decoder step, the recurrent GEMMs, the fused LSTM-cell
kernel, the query GEMM and the fused attention kernel (plus, with sample_prob > 0,
the step's projection and the scheduled-sampling kernel), and one output-projection
GEMM for all steps at the end.  Its gradient (nabu_speller_bwd) is the mirrored
loop with hand-written backward kernels; weight gradients that are sums over
steps are single GEMMs over all steps."""
import ctypes
from abc import ABCMeta, abstractmethod

import torch

from nabu_amd import _hip
from nabu_amd import variables as vs
from nabu_amd.autodiff import record, SeqLen
from nabu_amd.neuralnetworks.components import ops as nops
from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder


def _grad(var):
    if var.grad is None:
        var.grad = torch.zeros_like(var.data)
    return var.grad


def _ptrs(named, lstm, grad):
    """fill a nabu_speller_params / nabu_speller_grads struct"""
    s = _hip.SpellerPtrs()
    for k, var in named.items():
        setattr(s, k, None if var is None else _hip.ptr(_grad(var) if grad else var.data))
    for n, (kern, bias) in enumerate(lstm):
        s.lstm_kernel[n] = _hip.ptr(_grad(kern) if grad else kern.data)
        s.lstm_bias[n] = _hip.ptr(_grad(bias) if grad else bias.data)
    return s


def cell_parameters(cell, E):
    """The variables of a projected attention cell (created on first use in the current scope,
    TF-style names) in the form the C ABI takes them.  Returns (attention mechanism, LSTM cells,
    number of layers, num_units, output dim, {name: Variable}, [(kernel, bias) per layer])."""
    wrapper = cell._cell
    mech = wrapper.attention_mechanism
    cells = wrapper.cells
    nl = len(cells)
    U = cells[0].num_units
    if any(c.num_units != U for c in cells):
        raise NotImplementedError('all speller layers must have the same num_units')
    if nl > _hip.SPELLER_MAX_LAYERS:
        raise NotImplementedError('at most %d speller layers' % _hip.SPELLER_MAX_LAYERS)
    C = cell.output_size
    with vs.variable_scope('decoder'):
        av = mech.variables()
        with vs.variable_scope('attention_wrapper'):
            lstm = [c.variables(n, (C + E) if n == 0 else U) for n, c in enumerate(cells)]
        Wout, bout = cell.variables(E)
    named = dict(memory_kernel=av['memory_kernel'], query_kernel=av['query_kernel'],
                 attention_v=av['attention_v'], conv_kernel=av.get('conv_kernel'),
                 conv_proj=av.get('conv_proj'), out_kernel=Wout, out_bias=bout)
    return mech, cells, nl, U, C, named, lstm


def dynamic_decode(cell, encoded, encoded_seq_length, targets, target_seq_length, sample_prob,
                   is_training):  # noqa: C901
    """Run the projected attention cell over the target sequence.

    encoded [B,Te,E] (rows >= length zero), targets [B,Lt] int32 (already holding EOS
    where the recipe uses it), target_seq_length [B].  Returns logits [B,L,C] with
    L = max(target_seq_length); rows of finished utterances are zero."""
    dev = encoded.device
    B, Te, E = encoded.shape
    mech, cells, nl, U, C, named, lstm = cell_parameters(cell, E)
    tlen = SeqLen.wrap(target_seq_length, dev)
    elen = SeqLen.wrap(encoded_seq_length, dev)
    L = tlen.max()

    keep = cells[0].output_keep_prob
    seed, offset = nops.global_rng().next() if keep < 1 else (0, 0)
    if keep < 1:
        nops.global_rng().offset += L * nl          # one mask per (step, layer)
    # scheduled sampling (ScheduledEmbeddingTrainingHelper, rnn_decoder.py:59-66); like the
    # reference it is active whenever _decode runs, in the training and the validation graph
    sprob = float(sample_prob)
    sseed, soffset = nops.global_rng().next() if sprob > 0 else (0, 0)
    if sprob > 0:
        nops.global_rng().offset += L
    desc = _hip.SpellerDesc(ctypes.sizeof(_hip.SpellerDesc), B, Te, E, U, C, L, nl, mech.kind,
                            mech.filtersize, mech.numfilt, mech.prob_fn, keep, seed, offset * 1000003,
                            sprob, sseed, soffset * 1000003)
    lib = _hip.lib()
    reserve_bytes = lib.nabu_speller_reserve_bytes(ctypes.byref(desc))
    ws_bytes = lib.nabu_speller_ws_bytes(ctypes.byref(desc))
    if reserve_bytes == 0:
        raise _hip.NabuHipError('speller: unsupported shape: %s' % lib.nabu_last_error().decode())

    # decoder inputs: [SOS = C-1, y_0 .. y_{L-2}] (rnn_decoder.py:46-47), time-major ids
    ids = torch.full((L, B), C - 1, dtype=torch.int32, device=dev)
    if L > 1:
        ids[1:] = targets.to(torch.int32)[:, :L - 1].t()
    values = encoded if encoded.is_contiguous() else encoded.contiguous()
    logits = torch.empty((B, L, C), dtype=torch.float32, device=dev)
    reserve = torch.empty(reserve_bytes, dtype=torch.uint8, device=dev)
    ws = _hip.Workspace.get(ws_bytes, dev, 'speller')
    params = _ptrs(named, lstm, grad=False)
    ev = dynamic_decode.events          # (bench.py: a list while a step's decoder calls are to be bracketed by events)
    shape = dict(B=B, Te=Te, E=E, U=U, C=C, L=L)
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _hip.check(lib.nabu_speller_fwd(ctypes.byref(desc), _hip.ptr(values), _hip.ptr(elen.dev), _hip.ptr(ids),
                                    _hip.ptr(tlen.dev), ctypes.byref(params), _hip.ptr(logits),
                                    _hip.ptr(reserve), _hip.ptr(ws), ws_bytes, _hip.stream()),
               'nabu_speller_fwd')
    if ev is not None:
        e1.record()
        ev.append(('fwd', shape, e0, e1))

    def backward(dlogits):
        grads = _ptrs(named, lstm, grad=True)
        p2 = _ptrs(named, lstm, grad=False)
        dvalues = torch.empty_like(values)
        w2 = _hip.Workspace.get(ws_bytes, dev, 'speller')
        evb = dynamic_decode.events
        if evb is not None:
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
        _hip.check(lib.nabu_speller_bwd(ctypes.byref(desc), _hip.ptr(values), _hip.ptr(elen.dev),
                                        _hip.ptr(ids), _hip.ptr(tlen.dev), ctypes.byref(p2),
                                        _hip.ptr(dlogits.contiguous()), _hip.ptr(reserve),
                                        ctypes.byref(grads), _hip.ptr(dvalues), _hip.ptr(w2), ws_bytes,
                                        _hip.stream()), 'nabu_speller_bwd')
        if evb is not None:
            b1.record()
            evb.append(('bwd', shape, b0, b1))
        return [dvalues]

    record([encoded], [logits], backward)
    dynamic_decode.last = (desc, reserve)           # for decoder_inputs() below (tests, diagnostics)
    dynamic_decode.last_paths = (lib.nabu_speller_uses_persistent(ctypes.byref(desc), 0),
                                 lib.nabu_speller_uses_persistent(ctypes.byref(desc), 1))
    return logits, tlen



dynamic_decode.events = None


def beam_search(cell, encoded, encoded_seq_length, beam_width, max_steps, length_penalty=0.0,
                temperature=1.0, with_alignments=True):
    """Beam search over the projected attention cell (components/beam_search_decoder.py:68-451
    under dynamic_decode): ONE call into the C ABI (nabu_speller_beam_search), whose C++ driver runs
    the cell kernels on B*beam_width rows, prunes and gathers on the device, and stops as the
    reference's dynamic_decode does.  encoded [B,Te,E] (rows >= length zero).
    Returns (sequences [B,W,time] int32, lengths [B,W] int32, scores [B,W], alignments
    [B,W,time,Te] or None)."""
    dev = encoded.device
    B, Te, E = encoded.shape
    mech, cells, nl, U, C, named, lstm = cell_parameters(cell, E)
    elen = SeqLen.wrap(encoded_seq_length, dev)
    W, S = int(beam_width), int(max_steps)
    desc = _hip.BeamDesc(ctypes.sizeof(_hip.BeamDesc), B, Te, E, U, C, nl, mech.kind, mech.filtersize,
                         mech.numfilt, mech.prob_fn, W, S, float(length_penalty), float(temperature))
    lib = _hip.lib()
    ws_bytes = lib.nabu_speller_beam_ws_bytes(ctypes.byref(desc))
    if ws_bytes == 0:
        raise _hip.NabuHipError('beam search: unsupported shape: %s' % lib.nabu_last_error().decode())
    values = encoded if encoded.is_contiguous() else encoded.contiguous()
    seq = torch.empty((B, W, S), dtype=torch.int32, device=dev)
    lengths = torch.empty((B, W), dtype=torch.int32, device=dev)
    scores = torch.empty((B, W), dtype=torch.float32, device=dev)
    align = torch.empty((B, W, S, Te), dtype=torch.float32, device=dev) if with_alignments else None
    ws = _hip.Workspace.get(ws_bytes, dev, 'beam_search')
    params = _ptrs(named, lstm, grad=False)
    steps = ctypes.c_int32(0)
    _hip.check(lib.nabu_speller_beam_search(ctypes.byref(desc), _hip.ptr(values), _hip.ptr(elen.dev),
                                            ctypes.byref(params), _hip.ptr(seq), _hip.ptr(lengths),
                                            _hip.ptr(scores), _hip.ptr(align), ctypes.byref(steps),
                                            _hip.ptr(ws), ws_bytes, _hip.stream()), 'nabu_speller_beam_search')
    n = steps.value
    return seq[:, :, :n], lengths, scores, (align[:, :, :n] if with_alignments else None)

def decoder_inputs():
    """[L,B] int32 labels the last dynamic_decode fed to the cell (row 0 = SOS; later rows are the
    targets shifted by one, or samples where scheduled sampling replaced them)"""
    desc, reserve = dynamic_decode.last
    out = torch.empty((desc.L, desc.B), dtype=torch.int32, device=reserve.device)
    _hip.check(_hip.lib().nabu_speller_decoder_inputs(ctypes.byref(desc), _hip.ptr(reserve), _hip.ptr(out),
                                                      _hip.stream()), 'nabu_speller_decoder_inputs')
    return out


class RNNDecoder(ed_decoder.EDDecoder, metaclass=ABCMeta):
    '''a recurrent decoder: _decode runs the cell built by create_cell over the targets'''

    def _decode(self, encoded, encoded_seq_length, targets, target_seq_length, is_training):
        # only the first target / output is used (rnn_decoder.py:40-47)
        tname = list(targets.keys())[0]
        output_name = list(self.output_dims.keys())[0]
        rnn_cell = self.create_cell(encoded, encoded_seq_length, is_training)
        ename = list(encoded.keys())[0]
        logits, logit_seq_length = dynamic_decode(
            rnn_cell, encoded[ename], encoded_seq_length[ename], targets[tname],
            target_seq_length[tname], float(self.conf['sample_prob']), is_training)
        return {output_name: logits}, {output_name: logit_seq_length}, ()

    @abstractmethod
    def create_cell(self, encoded, encoded_seq_length, is_training):
        '''create the rnn cell'''

    def zero_state(self, encoded_dim, batch_size):
        '''the decoder zero state: zero cell/hidden states, context and alignments'''
        return ()
