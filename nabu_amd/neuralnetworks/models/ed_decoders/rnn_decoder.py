"""General recurrent decoder (reference: models/ed_decoders/rnn_decoder.py:13-127).

``_decode`` keeps the reference's structure — pad the targets with the start
label C-1, build the cell, run it over the whole target sequence — but
``tf.contrib.seq2seq.dynamic_decode`` over a ScheduledEmbeddingTrainingHelper is
``dynamic_decode`` below: a host loop that launches, per decoder step, the
recurrent GEMMs, the fused LSTM-cell kernel, the query GEMM and the fused
attention kernel, and one output-projection GEMM for all steps at the end.  Its
gradient is the mirrored loop with hand-written backward kernels; weight gradients
that are sums over steps are deferred to single GEMMs over all steps."""
from abc import ABCMeta, abstractmethod

import numpy as np
import torch

from nabu_amd import ops as hip
from nabu_amd import variables as vs
from nabu_amd.autodiff import record, SeqLen
from nabu_amd.neuralnetworks.components import ops as nops
from nabu_amd.neuralnetworks.models.ed_decoders import ed_decoder


def _zeros(*shape, device=None, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=device)


def _grad(var):
    if var.grad is None:
        var.grad = torch.zeros_like(var.data)
    return var.grad


def dynamic_decode(cell, encoded, encoded_seq_length, targets, target_seq_length, sample_prob,
                   is_training):
    """Run the projected attention cell over the target sequence.

    encoded [B,Te,E] (rows >= length zero), targets [B,Lt] int32 (already holding EOS
    where the recipe uses it), target_seq_length [B].  Returns logits [B,L,C] with
    L = max(target_seq_length); rows of finished utterances are zero."""
    if sample_prob > 0 and is_training:
        raise NotImplementedError(
            'scheduled sampling (sample_prob > 0) is not implemented on the HIP path yet; set '
            'sample_prob = 0 in the [decoder] section')
    wrapper = cell._cell
    mech = wrapper.attention_mechanism
    cells = wrapper.cells
    nl = len(cells)
    U = cells[0].num_units
    if any(c.num_units != U for c in cells):
        raise NotImplementedError('all speller layers must have the same num_units')
    dev = encoded.device
    B, Te, E = encoded.shape
    C = cell.output_size
    tlen = SeqLen.wrap(target_seq_length, dev)
    elen = SeqLen.wrap(encoded_seq_length, dev)
    L = tlen.max()
    F, K = mech.numfilt, mech.filtersize

    # ---- variables (created in the decoder's scope, TF-style names) -------------
    with vs.variable_scope('decoder'):
        av = mech.variables()
        with vs.variable_scope('attention_wrapper'):
            lstm = [c.variables(n, (C + E) if n == 0 else U) for n, c in enumerate(cells)]
        Wout, bout = cell.variables(E)
    Wmem, Wq, vatt = av['memory_kernel'], av['query_kernel'], av['attention_v']
    ck, wf = av.get('conv_kernel'), av.get('conv_proj')

    # ---- per-batch precomputation: keys = memory_layer(values) -------------------
    values = encoded if encoded.is_contiguous() else encoded.contiguous()
    keys = torch.empty((B, Te, U), dtype=torch.float32, device=dev)
    hip.gemm(values.view(B * Te, E), Wmem.data, keys.view(B * Te, U))
    desc = hip.attn_desc(B, Te, E, U, mech.kind, K, F)

    # decoder inputs: [SOS = C-1, y_0 .. y_{L-2}] (rnn_decoder.py:46-47), time-major ids
    tg = targets.to(torch.int32)
    ids = torch.full((L, B), C - 1, dtype=torch.int32, device=dev)
    if L > 1:
        ids[1:] = tg[:, :L - 1].t()
    ids = ids.contiguous()

    # ---- state, time-major; index t+1 holds the value after step t ----------------
    H = [_zeros(L + 1, B, U, device=dev) for _ in range(nl)]
    Cs = [_zeros(L + 1, B, U, device=dev) for _ in range(nl)]
    acts = [torch.empty((L, B, 4 * U), dtype=torch.float32, device=dev) for _ in range(nl)]
    ctx = _zeros(L + 1, B, E, device=dev)
    align = _zeros(L + 1, B, Te, device=dev)
    q_all = torch.empty((L, B, U), dtype=torch.float32, device=dev)
    z = torch.empty((B, 4 * U), dtype=torch.float32, device=dev)
    # outputs handed to the next layer / the attention (differs from H only with dropout)
    drop = [c.output_keep_prob for c in cells]
    Hout = [(_zeros(L + 1, B, U, device=dev) if drop[n] < 1 else H[n]) for n in range(nl)]
    seeds = {}

    for t in range(L):
        for n in range(nl):
            Kn, bn = lstm[n]
            if n == 0:
                hip.gemm(ctx[t], Kn.data[C:C + E], z)                       # previous context
                hip.gemm(H[0][t], Kn.data[C + E:], z, beta=1.0)            # previous h
                hip.lstm_cell_fwd(t, tlen.dev, z, bn.data, Kn.data[:C], ids[t], Cs[0][t], H[0][t],
                                  acts[0][t], Cs[0][t + 1], H[0][t + 1])
            else:
                hip.gemm(Hout[n - 1][t + 1], Kn.data[:U], z)
                hip.gemm(H[n][t], Kn.data[U:], z, beta=1.0)
                hip.lstm_cell_fwd(t, tlen.dev, z, bn.data, None, None, Cs[n][t], H[n][t],
                                  acts[n][t], Cs[n][t + 1], H[n][t + 1])
            if drop[n] < 1:
                seeds[(t, n)] = nops.global_rng().next()
                Hout[n][t + 1] = hip.dropout(H[n][t + 1], drop[n], *seeds[(t, n)])
        hip.gemm(Hout[-1][t + 1], Wq.data, q_all[t])                        # query_layer
        hip.attn_fwd(desc, t, tlen.dev, elen.dev, keys, values, q_all[t], vatt.data,
                     ck.data if ck is not None else None, wf.data if wf is not None else None,
                     align[t], ctx[t], align[t + 1], ctx[t + 1])

    # ---- output projection for all steps at once: [h_t, ctx_t] · W + b -----------------
    logits_tm = torch.empty((L, B, C), dtype=torch.float32, device=dev)
    hip.gemm(Hout[-1][1:].view(L * B, U), Wout.data[:U], logits_tm.view(L * B, C), bias=bout.data)
    hip.gemm(ctx[1:].view(L * B, E), Wout.data[U:], logits_tm.view(L * B, C), beta=1.0)
    logits = hip.swap01(logits_tm)
    hip.mask_time_(logits, tlen.dev)                                         # impute_finished

    def backward(dlogits):
        dl = hip.swap01(_as_lbf(dlogits, B, L, C))                           # [L,B,C]
        dl2 = dl.view(L * B, C)
        htop = Hout[-1][1:].view(L * B, U)
        hip.gemm(htop, dl2, _grad(Wout)[:U], trans_a=True)
        hip.gemm(ctx[1:].view(L * B, E), dl2, _grad(Wout)[U:], trans_a=True)
        hip.colsum(dl2, _grad(bout))
        dH = torch.empty((L, B, U), dtype=torch.float32, device=dev)         # d loss / d h_top[t] (direct)
        dCtx = torch.empty((L, B, E), dtype=torch.float32, device=dev)
        hip.gemm(dl2, Wout.data[:U], dH.view(L * B, U), trans_b=True)
        hip.gemm(dl2, Wout.data[U:], dCtx.view(L * B, E), trans_b=True)

        dkeys = _zeros(B, Te, U, device=dev)
        dv_part = _zeros(B, U, device=dev)
        dwf_part = _zeros(B, F, U, device=dev) if mech.kind else None
        dck_part = _zeros(B, K, F, device=dev) if mech.kind else None
        dq_all = torch.empty((L, B, U), dtype=torch.float32, device=dev)
        dz_all = [torch.empty((L, B, 4 * U), dtype=torch.float32, device=dev) for _ in range(nl)]
        dh_carry = [_zeros(B, U, device=dev) for _ in range(nl)]
        dh_next = [torch.empty((B, U), dtype=torch.float32, device=dev) for _ in range(nl)]
        dc_carry = [_zeros(B, U, device=dev) for _ in range(nl)]
        dc_next = [torch.empty((B, U), dtype=torch.float32, device=dev) for _ in range(nl)]
        dctx_carry = None
        dal_carry = None
        dal_buf = [_zeros(B, Te, device=dev), _zeros(B, Te, device=dev)] if mech.kind else None
        dx = torch.empty((B, U), dtype=torch.float32, device=dev)
        dctx_next = [torch.empty((B, E), dtype=torch.float32, device=dev) for _ in range(2)]

        for t in range(L - 1, -1, -1):
            if dctx_carry is not None:
                hip.axpy_(dCtx[t], dctx_carry)                               # context feeds step t+1's cell
            dal_out = dal_buf[t & 1] if mech.kind else None
            hip.attn_bwd(desc, t, tlen.dev, elen.dev, keys, values, q_all[t], vatt.data,
                         ck.data if ck is not None else None, wf.data if wf is not None else None,
                         align[t], align[t + 1], dCtx[t], dal_carry, dq_all[t], dkeys, dv_part,
                         dwf_part, dck_part, dal_out)
            dal_carry = dal_out
            hip.gemm(dq_all[t], Wq.data, dH[t], trans_b=True, beta=1.0)      # through the query layer
            dtop = dH[t]
            for n in range(nl - 1, -1, -1):
                Kn = lstm[n][0]
                dh_in = dtop
                if drop[n] < 1:
                    dh_in = hip.dropout(dh_in.contiguous(), drop[n], *seeds[(t, n)])
                hip.lstm_cell_bwd(t, tlen.dev, acts[n][t], Cs[n][t + 1], Cs[n][t], dh_in, dh_carry[n],
                                  dc_carry[n], dz_all[n][t], dc_next[n])
                dc_carry[n], dc_next[n] = dc_next[n], dc_carry[n]
                if n == 0:
                    nxt = dctx_next[t & 1]
                    hip.gemm(dz_all[0][t], Kn.data[C:C + E], nxt, trans_b=True)
                    dctx_carry = nxt
                    hip.gemm(dz_all[0][t], Kn.data[C + E:], dh_next[0], trans_b=True)
                else:
                    hip.gemm(dz_all[n][t], Kn.data[:U], dx, trans_b=True)
                    hip.gemm(dz_all[n][t], Kn.data[U:], dh_next[n], trans_b=True)
                    dtop = dx
                dh_carry[n], dh_next[n] = dh_next[n], dh_carry[n]

        # ---- deferred sums over steps ------------------------------------------------
        hip.gemm(htop, dq_all.view(L * B, U), _grad(Wq), trans_a=True)
        for n in range(nl):
            Kn, bn = lstm[n]
            dzn = dz_all[n].view(L * B, 4 * U)
            gK = _grad(Kn)
            if n == 0:
                hip.scatter_rows(ids.view(L * B), dzn, gK[:C])
                hip.gemm(ctx[:L].view(L * B, E), dzn, gK[C:C + E], trans_a=True)
                hip.gemm(H[0][:L].view(L * B, U), dzn, gK[C + E:], trans_a=True)
            else:
                hip.gemm(Hout[n - 1][1:].view(L * B, U), dzn, gK[:U], trans_a=True)
                hip.gemm(H[n][:L].view(L * B, U), dzn, gK[U:], trans_a=True)
            hip.colsum(dzn, _grad(bn))
        hip.colsum(dv_part, _grad(vatt))
        if mech.kind:
            hip.colsum(dwf_part.view(B, F * U), _grad(wf).view(F * U))
            hip.colsum(dck_part.view(B, K * F), _grad(ck).view(K * F))
        # keys = values · Wmem
        hip.gemm(values.view(B * Te, E), dkeys.view(B * Te, U), _grad(Wmem), trans_a=True)
        dvalues = torch.empty((B, Te, E), dtype=torch.float32, device=dev)
        hip.gemm(dkeys.view(B * Te, U), Wmem.data, dvalues.view(B * Te, E), trans_b=True)
        # context_t = align_t^T values: d values[b] += sum_t align_t[b] (x) dctx_t[b]
        al = align[1:]
        for b in range(B):
            hip.gemm(al[:, b], dCtx[:, b], dvalues[b], trans_a=True, beta=1.0,
                     M=Te, N=E, K=L, lda=B * Te, ldb=B * E, ldc=E)
        return [dvalues]

    record([encoded], [logits], backward)
    return logits, tlen


def _as_lbf(x, B, L, C):
    """[B,L,C] batch-major gradient viewed as the [L', B', C] input of swap01 that yields
    the time-major [L,B,C] tensor (swap01 maps x[l,b,:] -> y[b,l,:] for any (l,b) roles)."""
    return x.contiguous().view(B, L, C)


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
