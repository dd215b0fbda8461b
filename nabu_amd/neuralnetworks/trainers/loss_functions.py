"""Training losses (reference: nabu/neuralnetworks/trainers/loss_functions.py).

Each loss is one fused HIP kernel that produces the per-utterance loss AND the
gradient w.r.t. the logits in the same pass (as tf.nn.ctc_loss does); the
gradient is handed to the tape."""
import torch

from nabu_amd import ops as hip
from nabu_amd.autodiff import record, SeqLen, UNIT
from nabu_amd.neuralnetworks.components import ops

# device status words of the CTC kernels launched since the last check
pending_status = []


def factory(loss_function):
    '''get a callable loss(targets, logits, logit_seq_length, target_seq_length)
    (reference loss_functions.py:7-28)'''
    if loss_function == 'average_cross_entropy':
        return average_cross_entropy
    elif loss_function == 'CTC':
        return CTC
    elif loss_function == 'sum_cross_entropy':
        return sum_cross_entropy
    elif loss_function in ('average_sigmoid_cross_entropy', 'marigin'):
        raise Exception('loss function %s is outside the MI355X hot path' % loss_function)
    else:
        raise Exception('unknown loss function %s' % loss_function)


def _labels(t):
    if t.dtype != torch.int32 or not t.is_contiguous():
        t = t.to(torch.int32).contiguous()
    return t


def _total(losses):
    '''sum over the outputs (reference loss_functions.py:212: tf.reduce_sum over the per-output
    losses).  With several outputs the sum is a node of its own and is recorded on the tape, so that
    the backward pass reaches every term (each gets the unit gradient).'''
    if len(losses) == 1:
        return losses[0]
    total = losses[0].clone()
    for l in losses[1:]:
        hip.axpy_(total, l)
    record(losses, [total], lambda g: [UNIT] * len(losses))
    return total


def CTC(targets, logits, logit_seq_length, target_seq_length):
    '''CTC loss (reference loss_functions.py:180-214): mean over the batch of
    -log p(targets | logits), summed over the outputs; blank = last class.'''
    losses = []
    for t in targets:
        lg = logits[t]
        B = lg.shape[0]
        lsl, tsl = SeqLen.wrap(logit_seq_length[t], lg.device), SeqLen.wrap(target_seq_length[t], lg.device)
        labels, _ = ops.dense_sequence_to_sparse(_labels(targets[t]), tsl)
        nll, dlogits, status = hip.ctc_loss_grad(lg.contiguous(), lsl.dev, labels, tsl.dev, 1.0 / B)
        pending_status.append(status)
        loss = hip.sum_(nll, 1.0 / B)
        record([lg], [loss], lambda g, d=dlogits: [d])
        losses.append(loss)
    return _total(losses)


def average_cross_entropy(targets, logits, logit_seq_length, target_seq_length):
    '''cross entropy averaged over timesteps (reference loss_functions.py:155-165):
    mean_b( sum_{t<logit_len} xent / target_len ), summed over the outputs.'''
    losses = []
    for t in targets:
        lg = logits[t]
        B = lg.shape[0]
        lsl, tsl = SeqLen.wrap(logit_seq_length[t], lg.device), SeqLen.wrap(target_seq_length[t], lg.device)
        per_utt, dlogits = hip.xent_loss_grad(lg.contiguous(), _labels(targets[t]), lsl.dev, tsl.dev, 1.0 / B)
        loss = hip.sum_(per_utt, 1.0 / B)
        record([lg], [loss], lambda g, d=dlogits: [d])
        losses.append(loss)
    return _total(losses)


def sum_cross_entropy(targets, logits, logit_seq_length, target_seq_length):
    '''cross entropy summed over timesteps (reference loss_functions.py:142-153): the mask is the
    TARGET length here and nothing is divided: mean_b( sum_{t<target_len} xent ), summed over the
    outputs.  Same kernel as average_cross_entropy with a divisor of one.'''
    losses = []
    for t in targets:
        lg = logits[t]
        B = lg.shape[0]
        tsl = SeqLen.wrap(target_seq_length[t], lg.device)
        ones = torch.ones_like(tsl.dev)
        per_utt, dlogits = hip.xent_loss_grad(lg.contiguous(), _labels(targets[t]), tsl.dev, ones, 1.0 / B)
        loss = hip.sum_(per_utt, 1.0 / B)
        record([lg], [loss], lambda g, d=dlogits: [d])
        losses.append(loss)
    return _total(losses)


def check_status():
    '''raise (as tf.nn.ctc_loss does) if any CTC utterance had no valid alignment'''
    global pending_status
    todo, pending_status = pending_status, []
    for s in todo:
        code = int(s.item())
        if code:
            raise Exception('CTC: Not enough time for target transition sequence '
                            '(utterance %d of the batch)' % (code - 1))
    # a persistent recurrent kernel that gave up (bounded-spin timeout) leaves the results of the
    # step invalid and its status word set: raise here, where the training / validation loops
    # already synchronise
    hip.check_persist_status()
